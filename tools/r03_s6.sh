#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_adversarial.py -q > $O/s6_adv.log 2>&1; tail -30 $O/s6_adv.log
