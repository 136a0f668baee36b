#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python3 tools/gpu_owntree_cmp.py > $O/s9_cmp.log 2>&1; cat $O/s9_cmp.log
