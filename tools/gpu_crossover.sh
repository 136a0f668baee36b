#!/bin/bash
# Where does MPT_PIPE_AUTO's switch from k_wavelocal to k_ordered belong?  Scenes of 1, 2, 3, 5, 8 bunnies (4,968 triangles each, the
# camera of scene.xml on the first rows of bunny20.xml), device-built trees with the leaf size each kernel wants, one serial render each.
D=$(mktemp -d /tmp/mpt_cross_XXXX); cp assets/bunny.obj $D/
for N in 1 2 3 5 8; do
  { echo "<Scene>"; sed -n 3,4p assets/bunny20.xml; grep "<Mesh" assets/bunny20.xml | head -$N; echo "</Scene>"; } > $D/b$N.xml
  for cfg in "2 6" "3 2" "3 4"; do
    set -- $cfg
    SCENE=$D/b$N.xml BVH=3 PIPE=$1 MPT_LBVH_LEAF=$2 SPP=${SPP:-128} REPS=4 python3 tools/prof_run.py > $D/log 2>&1 || { echo "b$N pipe $1 leaf $2: FAILED"; tail -2 $D/log; continue; }
    python3 - $D/log "bunnies $N pipe $1 leaf $2" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
rows = re.findall(r"total_ms ([\d.]+) trace_ms ([\d.]+) launches \d+ rays (\d+)", t)[1:]
prims = re.search(r'"prims": (\d+)', t).group(1)
ms = min(float(r[0]) for r in rows)
print("%s (%s primitives): %.2f ms  %.2f Grays/s" % (sys.argv[2], prims, ms, int(rows[0][2]) / ms / 1e6))
PY
  done
done
rm -rf $D
