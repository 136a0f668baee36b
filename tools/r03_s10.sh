#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 800 python3 -m pytest tests/test_gpu_lbvh.py -x -q > $O/s10_lbvh.log 2>&1; tail -8 $O/s10_lbvh.log
timeout -k 10 600 python3 tools/gpu_owntree_cmp.py > $O/s10_cmp.log 2>&1; cat $O/s10_cmp.log
timeout -k 10 600 python3 tools/gpu_devbuild.py 64 > $O/s10_devbuild.log 2>&1; cat $O/s10_devbuild.log
