#!/bin/bash
# Refreshes the rocprofv3 evidence for a round on a GPU box:  bash tools/profile_round.sh r01
#   1. rocprofv3 --kernel-trace --stats of the default `python3 bench.py` run  -> gpurun_out/<tag>_kernel_stats.csv
#      (+ the bench line printed under the profiler)                            -> gpurun_out/<tag>_bench_under_rocprof.json
#   2. three --pmc passes (own runs, no tracing) of tools/prof_run.py, 256 spp  -> gpurun_out/<tag>_pmc_*.txt
# Copy what should be judged from gpurun_out/ into profiles/ afterwards (tools/pmc_to_json.py builds the JSON).
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_stats
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o st -- python3 $ROOT/bench.py > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_bench_under_rocprof.err
cp $(ls $OUT/prof_stats/*kernel_stats.csv $OUT/prof_stats/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_kernel_stats.csv
tail -1 $OUT/${TAG}_bench_under_rocprof.json | cut -c1-400
head -6 $OUT/${TAG}_kernel_stats.csv
export SPP=256 PIPE=2 REPS=2
# one counter set per pass: FETCH_SIZE + WRITE_SIZE together exceed what the hardware collects at once (rocprofiler
# aborts and the process hangs), so every pass gets its own timeout
pass() {  # pass <name> <counters...>
    local name=$1; shift
    echo "pmc pass $name: $*"
    cd /tmp
    timeout -k 5 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o $name -- python3 $ROOT/tools/prof_run.py > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed (see gpurun_out/pmc_$name.log)"
    cd $ROOT
}
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_hit $OUT/pmc_miss $OUT/pmc_sq
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass hit TCC_HIT_sum
pass miss TCC_MISS_sum
pass sq SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
cd $ROOT
python3 tools/pmc_parse.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_hit $OUT/pmc_miss $OUT/pmc_sq | tee $OUT/${TAG}_pmc_raw.txt
