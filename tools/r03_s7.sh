#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python3 tools/gpu_devbuild.py 64 > $O/s7_devbuild.log 2>&1; cat $O/s7_devbuild.log
