#!/bin/bash
# VALU instructions per ray at depth 1 / 2 / 8 (what does a primary ray cost, what does a bounce ray cost)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export SPP=${SPP:-128} PIPE=2 REPS=2
for D in 1 2 8; do
  export DEPTH=$D
  rm -rf $OUT/pmc_d$D
  cd /tmp && export TMPDIR=/tmp
  timeout -k 5 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_d$D -o d -- python3 $ROOT/tools/prof_run.py > $OUT/pmc_d$D.log 2>&1 || echo "depth $D failed"
  cd $ROOT
  grep "rays" $OUT/pmc_d$D.log | tail -1
  python3 tools/pmc_parse.py $OUT/pmc_d$D
done
