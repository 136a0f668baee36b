#!/bin/bash
# knob sweep of the closest-first pipeline on bunny x20 (scratch tool)
export SPP=256 PIPE=3 REPS=2 SCENE=bunny20.xml BVH=1
run() { echo -n "$*: "; env "$@" python3 tools/prof_run.py | tail -1; }
run X=0
run MPT_OT_BUDGETS=8
run MPT_OT_BUDGETS=12
run MPT_OT_BUDGETS=24
run MPT_OT_BUDGETS=32
run MPT_OT_INPLACE=24
run MPT_OT_INPLACE=32
run MPT_OT_INPLACE=40
run MPT_OT_INPLACE=56
run MPT_OT_INPLACE=65
run MPT_OT_MIN_ACTIVE=0,16
run MPT_OT_MIN_ACTIVE=0,32
run MPT_OT_STACK=6
run X=0
