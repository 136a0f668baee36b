#!/bin/bash
# knob sweep of the reference-order wave-local kernel (scratch tool)
export SPP=${SPP:-256} PIPE=2 REPS=3 SCENE=${SCENE:-scene.xml}
run() { echo -n "$*: "; env "$@" python3 tools/prof_run.py | tail -1; }
run X=0
run MPT_WL_MIN=128
run MPT_WL_MIN=256
run MPT_WL_BLOCK=2048
run MPT_WL_BLOCK=512
run MPT_TILE_ORDER=0
run MPT_TILE_ORDER=1
run MPT_TILE_ORDER=2
run MPT_WL_DIV=16
run MPT_WL_DIV=48
run MPT_BUDGETS=8,20,50,125
run MPT_BUDGETS=10,24,60,150
run MPT_MIN_ACTIVE=0,32,32,32,32
run X=0
