set -e
mkdir -p gpurun_out/r05
D=$(mktemp -d /tmp/mpt_occ_XXXX); cp assets/bunny.obj $D/
{ echo "<Scene>"; sed -n 3,4p assets/bunny20.xml; grep "<Mesh" assets/bunny20.xml | head -8; echo "</Scene>"; } > $D/b8.xml
V='base base:MPT_OT_OCC=5 base:MPT_OT_OCC=6 base:MPT_OT_BUDGETS=64+MPT_OT_MIN_ACTIVE=40,24 base:MPT_OT_BUDGETS=64+MPT_OT_MIN_ACTIVE=48,24 base:MPT_OT_BUDGETS=32+MPT_OT_MIN_ACTIVE=32,24 base:MPT_OT_MIN_ACTIVE=32,24 base:MPT_OT_MIN_ACTIVE=0,32 base'
SCENES="bunny20.xml $D/b8.xml" REPS=4 tools/gpu_ab.sh $V > gpurun_out/r05/s9_ab.log 2>&1
cat gpurun_out/r05/s9_ab.log
SCENES=config4 SHARDS=8 BSDF=1 DEPTH=16 SPP=4096 REPS=3 tools/gpu_ab.sh base base:MPT_OT_OCC=5 base:MPT_OT_OCC=6 base > gpurun_out/r05/s9_ab_c4.log 2>&1
cat gpurun_out/r05/s9_ab_c4.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05/s9_tests.log 2>&1 || true
tail -5 gpurun_out/r05/s9_tests.log
