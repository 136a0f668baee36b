#!/bin/bash
# round-3 session 1: baseline of the inherited build on this round's box + memory-pipe counters of k_ordered on bunny x20
O=gpurun_out; mkdir -p $O
python3 bench.py --steps 4 --warmup 1 > $O/s1_bench.json 2> $O/s1_bench.err && tail -c 600 $O/s1_bench.json
SCENE=bunny20.xml BVH=1 PIPE=3 SPP=256 REPS=3 python3 tools/prof_run.py > $O/s1_b20.log 2>&1; tail -2 $O/s1_b20.log
BVH=1 MPT_LIB=$PWD/metalpathtracer_amd/lib/libmpt_hip_times.so python3 tools/gpu_ot_times.py bunny20.xml 64 > $O/s1_times.log 2>&1; cat $O/s1_times.log
SCENE=bunny20.xml BVH=1 SPP=64 bash tools/pmc_mem.sh s1 k_ordered > $O/s1_mem.log 2>&1; tail -3 $O/s1_mem.log
