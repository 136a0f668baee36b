#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call:  bash tools/profiles_round.sh
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
PIPE=2 bash tools/pmc_round.sh r02wl k_wavelocal > $OUT/pmc_wl.log 2>&1 || exit 1
PIPE=3 bash tools/pmc_round.sh r02ot k_ordered > $OUT/pmc_ot.log 2>&1 || exit 1
SCENE=bunny20.xml BVH=1 SPP=64 PIPE=3 bash tools/pmc_round.sh r02otb k_ordered > $OUT/pmc_otb.log 2>&1 || exit 1
SCENE=bunny20.xml BVH=1 SPP=64 PIPE=2 bash tools/pmc_round.sh r02wlb k_wavelocal > $OUT/pmc_wlb.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/bench_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o b -- python3 $ROOT/bench.py --steps 8 --warmup 2 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
cd $ROOT
find $OUT/bench_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
head -5 $OUT/bench_kernel_stats.csv
