#!/bin/bash
# claim-size divisor across workloads (scratch tool)
run() { echo -n "$*: "; env "$@" python3 tools/prof_run.py | tail -1; }
for D in 16 32 48 64; do
  run MPT_WL_DIV=$D SCENE=scene.xml PIPE=2 SPP=256 REPS=3
  run MPT_WL_DIV=$D SCENE=scene.xml PIPE=2 SPP=32 REPS=3
  run MPT_WL_DIV=$D SCENE=scene.xml PIPE=2 SPP=256 REPS=3 W=640 H=360
  run MPT_WL_DIV=$D SCENE=glass.xml PIPE=2 SPP=256 REPS=3
  run MPT_WL_DIV=$D SCENE=bunny20.xml BVH=1 PIPE=3 SPP=256 REPS=2
  run MPT_WL_DIV=$D SCENE=bunny20.xml BVH=1 PIPE=3 SPP=32 REPS=2
done
