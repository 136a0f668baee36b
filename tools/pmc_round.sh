#!/bin/bash
# Counter evidence for the dominant kernel on a GPU box:  bash tools/pmc_round.sh <tag> [kernel-substring]
# One rocprofv3 --pmc pass per counter set (no tracing in the same run), each on tools/prof_run.py (scene.xml 1080p,
# SPP samples, PIPE pipeline, REPS launches; the LAST dispatch of the kernel is reported).  The sets respect the
# gfx950 slot limits of MI355X_MICROARCH.md (SQ 8, TCC: FETCH_SIZE and WRITE_SIZE in passes of their own, GRBM 2).
# Result: gpurun_out/<tag>_pmc.json (tools/pmc_collect.py) — copy it to profiles/ to have it judged.
TAG=${1:-r02}; KERNEL=${2:-k_wavelocal}
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export SPP=${SPP:-256} PIPE=${PIPE:-2} REPS=${REPS:-2}
SETS=(
 "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_SALU"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum"
)
dirs=""
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1)); D=$OUT/pmc_${TAG}_$i; rm -rf $D
  cd /tmp && export TMPDIR=/tmp
  if ! timeout -k 5 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $SET --output-format csv -d $D -o s -- python3 $ROOT/tools/prof_run.py > $D.log 2>&1; then
    echo "set $i ($SET) FAILED: $(grep -m1 -i 'error\|exceeds\|signal' $D.log | cut -c1-200)"; exit 1
  fi
  cd $ROOT
  dirs="$dirs $D"
  grep -h "Mrays" $D.log | tail -1
done
python3 tools/pmc_collect.py $KERNEL $OUT/${TAG}_pmc.json $dirs
