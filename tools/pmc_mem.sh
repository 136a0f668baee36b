#!/bin/bash
# Memory-pipe counters of one kernel (texture addresser, vector L1, texture data, L2 requests):  bash tools/pmc_mem.sh <tag> [kernel-substring]
# One rocprofv3 --pmc pass per counter set on tools/prof_run.py (env SCENE BVH SPP PIPE REPS as there), no tracing.
# Result: gpurun_out/<tag>_mem.json (tools/pmc_collect.py).
# Slot limits on gfx950: the texture addresser (TA) and the texture data unit (TD) take TWO counters per pass, the vector L1
# (TCP) and the L2 (TCC) four.  Round 3's sets asked for four TA_* and three TD_* counters at once: rocprofiler_create_counter_config
# answered error 38 ("Request exceeds the capabilities of the hardware to collect") and rocprofv3 aborted before any kernel ran —
# and an `|| echo` here hid it.  A pass that fails now FAILS THE SCRIPT (exit 1, nothing after it runs).
TAG=${1:-mem}; KERNEL=${2:-k_ordered}
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export SPP=${SPP:-64} PIPE=${PIPE:-3} REPS=${REPS:-1}
SETS=(
 "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum"
 "TD_LOAD_WAVEFRONT_sum"
 "TCC_REQ_sum TCC_READ_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
)
dirs=""
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1)); D=$OUT/pmcmem_${TAG}_$i; rm -rf $D
  cd /tmp && export TMPDIR=/tmp
  if ! timeout -k 5 240 rocprofv3 --pmc $SET --output-format csv -d $D -o s -- python3 $ROOT/tools/prof_run.py > $D.log 2>&1; then
    echo "set $i ($SET) FAILED: $(grep -m1 -i 'error\|exceeds\|signal' $D.log | cut -c1-200)"; exit 1
  fi
  cd $ROOT
  dirs="$dirs $D"
  grep -h "Mrays" $D.log | tail -1
done
python3 tools/pmc_collect.py $KERNEL $OUT/${TAG}_mem.json $dirs
