#!/bin/bash
# A/B of library variants: usage gpu_variants.sh "name:ENV=.. ENV=.." ...   (lib = metalpathtracer_amd/lib/libmpt_hip_<name>.so, "base" = the product)
export SPP=${SPP:-256} PIPE=${PIPE:-3} REPS=3
for SCENE in ${SCENES:-scene.xml bunny20.xml}; do
  export SCENE
  for v in "$@"; do
    n=${v%%:*}; e=${v#*:}; [ "$e" = "$v" ] && e=""
    lib=$PWD/metalpathtracer_amd/lib/libmpt_hip_$n.so; [ "$n" = base ] && lib=$PWD/metalpathtracer_amd/lib/libmpt_hip.so
    echo -n "$SCENE $n [$e]: "; env MPT_LIB=$lib $e python3 tools/prof_run.py | tail -1
  done
done
