#!/bin/bash
# Extra counter sets (instruction cache, instruction mix) for one kernel:  bash tools/pmc_extra.sh <tag> [kernel-substring]
TAG=${1:-x}; KERNEL=${2:-k_wavelocal}
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export SPP=${SPP:-256} PIPE=${PIPE:-2} REPS=${REPS:-2}
SETS=(
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE GRBM_GUI_ACTIVE"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU"
 "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL"
)
dirs=""
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1)); D=$OUT/pmcx_${TAG}_$i; rm -rf $D
  cd /tmp && export TMPDIR=/tmp
  timeout -k 5 240 rocprofv3 --pmc $SET --output-format csv -d $D -o s -- python3 $ROOT/tools/prof_run.py > $D.log 2>&1 \
    || echo "set $i failed: $(grep -m1 -i 'error\|exceeds' $D.log | cut -c1-200)"
  cd $ROOT
  dirs="$dirs $D"
  grep -h "Mrays" $D.log | tail -1
done
python3 tools/pmc_collect.py $KERNEL $OUT/${TAG}_pmcx.json $dirs
