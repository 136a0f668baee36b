#!/bin/bash
# issue-side counters of the dominant kernel at 256 spp (one pass per counter set)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export SPP=${SPP:-256} PIPE=2 REPS=2
i=0
for SET in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf $OUT/pmc_s$i
  cd /tmp && export TMPDIR=/tmp
  timeout -k 5 240 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_s$i -o s -- python3 $ROOT/tools/prof_run.py > $OUT/pmc_s$i.log 2>&1 || echo "set $i failed: $(grep -m1 -i 'error\|exceeds' $OUT/pmc_s$i.log | cut -c1-200)"
  cd $ROOT
  python3 tools/pmc_parse.py $OUT/pmc_s$i
done
grep rays $OUT/pmc_s1.log | tail -1
