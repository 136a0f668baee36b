#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 800 python3 -m pytest tests/test_gpu_lbvh.py -x -q > $O/s5_lbvh.log 2>&1; tail -25 $O/s5_lbvh.log
timeout -k 10 700 python3 -m pytest tests/test_gpu_adversarial.py -x -q > $O/s5_adv.log 2>&1; tail -15 $O/s5_adv.log
