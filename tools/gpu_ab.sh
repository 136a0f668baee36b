#!/bin/bash
# Same-box A/B of library builds and knobs, min and median of REPS serial renders each:
#   tools/gpu_ab.sh "name[:ENV=.. ENV=..]" ...      name = base (the product) or the <name> of metalpathtracer_amd/lib/libmpt_hip_<name>.so
# Workload from the environment: SCENES (default "scene.xml bunny20.xml"), SPP (256), BVH (3 = device build), PIPE (4 = AUTO), REPS (5).
export SPP=${SPP:-256} PIPE=${PIPE:-4} BVH=${BVH:-3} REPS=${REPS:-5}
for SCENE in ${SCENES:-scene.xml bunny20.xml}; do
  export SCENE
  for v in "$@"; do
    n=${v%%:*}; e=${v#*:}; [ "$e" = "$v" ] && e=""; e=${e//+/ }   # (several ENV entries: joined by +)
    lib=$PWD/metalpathtracer_amd/lib/libmpt_hip_$n.so; [ "$n" = base ] && lib=$PWD/metalpathtracer_amd/lib/libmpt_hip.so
    [ -f "$lib" ] || { echo "$SCENE $n: $lib missing"; continue; }
    env MPT_LIB=$lib $e python3 tools/prof_run.py > /tmp/ab_$$.log 2>&1 || { echo "$SCENE $n [$e]: FAILED $(tail -1 /tmp/ab_$$.log)"; continue; }
    TAG="$SCENE pipe=$PIPE bvh=$BVH spp=$SPP $n [$e]" python3 - /tmp/ab_$$.log <<'PY'
import os, re, sys
rows = re.findall(r"total_ms ([\d.]+) trace_ms ([\d.]+) launches \d+ rays (\d+)", open(sys.argv[1]).read())[1:]   # first render = warm-up
ms = sorted(float(r[0]) for r in rows)
print("%s: min %.2f ms  median %.2f ms  (%d reps after warm-up)  %.2f Grays/s at min" % (os.environ["TAG"], ms[0], ms[len(ms) // 2], len(ms), int(rows[0][2]) / ms[0] / 1e6))
PY
  done
done
rm -f /tmp/ab_$$.log
