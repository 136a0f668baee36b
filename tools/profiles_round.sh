#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call:  bash tools/profiles_round.sh [tag]      (tag: r03)
# Counter passes (one rocprofv3 --pmc run per counter set, no tracing), the kernel summary of bench.py, the diagnostics-build
# tables, the in-kernel clocks, the device-build and shard measurements.  Everything lands in gpurun_out/<tag>_*; run
# `python tools/profiles_collect.py <tag>` afterwards (here) to turn it into profiles/<tag>_*.
TAG=${1:-r05}; PART=${2:-all}    # part 1: counter passes + the bench lines; part 2: diagnostics tables, clocks, builds, configs (two GPU calls of <= 20 min)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; LIB=$ROOT/metalpathtracer_amd/lib
if [ "$PART" != 2 ]; then
# one counter profile per workload of the bench line (the headline step and its four extras) + the two cross runs of round 3
BVH=3 PIPE=2 ASYNC=1 bash tools/pmc_round.sh ${TAG}wl k_wavelocal > $OUT/${TAG}_pmc_wl.log 2>&1 || exit 1   # (the variant the timed steps run: k_wavelocal_corun)
BVH=0 PIPE=2 bash tools/pmc_round.sh ${TAG}wlref k_wavelocal > $OUT/${TAG}_pmc_wlref.log 2>&1 || exit 1
SCENE=cornell.xml CAM=cornell BVH=3 PIPE=2 bash tools/pmc_round.sh ${TAG}cor k_wavelocal > $OUT/${TAG}_pmc_cor.log 2>&1 || exit 1
SCENE=bunny20.xml BVH=3 SPP=256 PIPE=3 bash tools/pmc_round.sh ${TAG}otb k_ordered > $OUT/${TAG}_pmc_otb.log 2>&1 || exit 1
SCENE=config4 BVH=3 SPP=4096 SHARDS=8 BSDF=1 DEPTH=16 PIPE=3 PMC_TIMEOUT=400 bash tools/pmc_round.sh ${TAG}c4 k_ordered > $OUT/${TAG}_pmc_c4.log 2>&1 || exit 1
BVH=3 PIPE=3 bash tools/pmc_round.sh ${TAG}ot k_ordered > $OUT/${TAG}_pmc_ot.log 2>&1 || exit 1
SCENE=bunny20.xml BVH=3 SPP=64 PIPE=2 bash tools/pmc_round.sh ${TAG}wlb k_wavelocal > $OUT/${TAG}_pmc_wlb.log 2>&1 || exit 1
SCENE=bunny20.xml BVH=3 SPP=64 PIPE=3 bash tools/pmc_mem.sh ${TAG}otb k_ordered > $OUT/${TAG}_mem_otb.log 2>&1 || exit 1
python3 tools/profiles_collect.py $TAG --pmc-only || exit 1   # profiles/<tag>_pmc_*.json of THIS build, for the bench lines below
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/bench_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o b -- python3 $ROOT/bench.py --steps 8 --warmup 2 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_bench_under_rocprof.err || exit 1
cd $ROOT
find $OUT/bench_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_bench_kernel_stats.csv
python3 bench.py --steps 8 --warmup 2 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
head -c 600 $OUT/${TAG}_bench.json; echo
fi
[ "$PART" = 1 ] && exit 0
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/devb_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/devb_trace -o b -- python3 $ROOT/tools/prof_devbuild.py > $OUT/${TAG}_devbuild_prof.log 2>&1 || exit 1
cd $ROOT
find $OUT/devb_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_devbuild_kernel_stats.csv
python3 tools/trace_timeline.py $(find $OUT/devb_trace -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_devbuild_timeline.txt 2>&1 || exit 1
COUNT=1 SPP=128 JSON_OUT=$OUT/${TAG}_step_table.json MPT_LIB=$LIB/libmpt_hip_wavetimes.so timeout -k 10 300 python3 tools/gpu_wave_times.py > $OUT/${TAG}_step_table.txt 2>&1 || exit 1
BVH=3 JSON_OUT=$OUT/${TAG}_ot_times_bunny20.json MPT_LIB=$LIB/libmpt_hip_times.so timeout -k 10 300 python3 tools/gpu_ot_times.py bunny20.xml 64 > $OUT/${TAG}_ot_times_bunny20.txt 2>&1 || exit 1
BVH=3 MPT_LIB=$LIB/libmpt_hip_times.so timeout -k 10 300 python3 tools/gpu_ot_times.py config4 64 > $OUT/${TAG}_ot_times_config4.txt 2>&1 || exit 1
{ echo "Where k_ordered's node and primitive bytes come from, and what its stack pops cost (diagnostics build -DMPT_OT_TIMES, tools/gpu_ot_times.py, 64 spp; VERDICT r4 item 1a):"; for w in bunny20 config4; do head -1 $OUT/${TAG}_ot_times_$w.txt; grep -h "served from LDS\|stack pops\|lane utilisation" $OUT/${TAG}_ot_times_$w.txt; done; } > $OUT/${TAG}_ot_lds_share.txt
timeout -k 10 300 python3 tools/gpu_draw_fps.py > $OUT/${TAG}_draw_fps.txt 2>&1 || exit 1
MPT_LIB=$LIB/libmpt_hip_clock.so timeout -k 10 300 python3 tools/gpu_clock.py > $OUT/${TAG}_inkernel_clock.txt 2>&1 || exit 1
timeout -k 10 600 python3 tools/gpu_devbuild.py 64 > $OUT/${TAG}_devbuild.txt 2>&1 || exit 1
timeout -k 10 600 python3 tools/gpu_scene_digest.py --time --check tests/golden/devbuild_digests.json > $OUT/${TAG}_devbuild_digests.txt 2>&1 || exit 1
timeout -k 10 600 python3 tools/gpu_shard_time.py > $OUT/${TAG}_shard_time.txt 2>&1 || exit 1
timeout -k 10 900 python3 tools/gpu_depth_work.py > $OUT/${TAG}_depth_work.txt 2>&1 || exit 1
timeout -k 10 900 python3 tools/gpu_configs.py > $OUT/${TAG}_configs.txt 2>&1 || exit 1
tail -3 $OUT/${TAG}_configs.txt
