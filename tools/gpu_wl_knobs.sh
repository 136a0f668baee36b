#!/bin/bash
# knob sweep of the reference-order wave-local kernel on scene.xml (scratch tool)
export SPP=${SPP:-256} PIPE=2 REPS=3 SCENE=${SCENE:-scene.xml}
run() { echo -n "$*: "; env "$@" python3 tools/prof_run.py | tail -1; }
run X=0
run MPT_WL_DIV=24
run MPT_WL_DIV=32
run MPT_WL_DIV=48
run MPT_WL_DIV=64
run MPT_WL_DIV=128
run MPT_WL_DIV=256
run MPT_WL_DIV=32 MPT_WL_BLOCK=512
run MPT_WL_DIV=32 MPT_WL_BLOCK=256
run MPT_WL_DIV=64 MPT_WL_BLOCK=256
run MPT_WL_DIV=32 MPT_BUDGETS=10,24,60,150
run MPT_WL_DIV=32 MPT_MIN_ACTIVE=0,32,32,32,32
run MPT_WL_DIV=32 MPT_BUDGETS=10,24,60,150 MPT_MIN_ACTIVE=0,32,32,32,32
run X=0
