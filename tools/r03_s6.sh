#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 800 python3 -m pytest tests/test_gpu_lbvh.py -x -q > $O/s6_lbvh.log 2>&1; tail -25 $O/s6_lbvh.log
timeout -k 10 600 python3 tools/gpu_devbuild.py 64 > $O/s6_devbuild.log 2>&1; cat $O/s6_devbuild.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_adversarial.py -q > $O/s6_adv.log 2>&1; tail -30 $O/s6_adv.log
