#!/bin/bash
# quick knob sweep of the closest-first pipeline on scene.xml / bunny20.xml 1080p (scratch tool)
export SPP=${SPP:-256} PIPE=3 REPS=3
for SCENE in scene.xml bunny20.xml; do
  export SCENE
  for B in "2,6" "4,10" "6,16" "8,20" "12,32" "4,1000000" "1000000,1000000"; do echo -n "$SCENE budgets $B: "; MPT_OT_BUDGETS=$B python3 tools/prof_run.py | tail -1; done
  for A in "0,0,0" "0,16,16" "0,32,32" "16,24,24"; do echo -n "$SCENE min_active $A: "; MPT_OT_MIN_ACTIVE=$A python3 tools/prof_run.py | tail -1; done
done
