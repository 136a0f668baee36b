#!/bin/bash
# quick knob sweep of the closest-first pipeline on scene.xml / bunny20.xml 1080p (scratch tool)
export SPP=${SPP:-256} PIPE=3 REPS=3
for SCENE in scene.xml bunny20.xml; do
  export SCENE
  for W in 1 8 16 24 32 48 65; do echo -n "$SCENE walk_now $W: "; MPT_OT_WALK_NOW=$W python3 tools/prof_run.py | tail -1; done
  for S in 4 6 8 10; do echo -n "$SCENE stack $S: "; MPT_OT_STACK=$S python3 tools/prof_run.py | tail -1; done
done
