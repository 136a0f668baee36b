#!/bin/bash
# MPT_OT_INPLACE sweep (scratch tool)
export SPP=256 PIPE=3 REPS=2
for SC in "bunny20.xml 1" "scene.xml 0"; do
  set -- $SC; export SCENE=$1 BVH=$2
  for T in 65 56 48 40 32 24 65; do
    echo -n "$SCENE inplace $T: "; MPT_OT_INPLACE=$T python3 tools/prof_run.py | tail -1
  done
done
