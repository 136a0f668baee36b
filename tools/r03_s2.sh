#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ordered.py -x -q -k "bit_exact or full_size or in_place" > $O/s2_ord.log 2>&1; tail -5 $O/s2_ord.log
BVH=1 SCENES="bunny20.xml" bash tools/gpu_variants.sh base7:MPT_OT_REFILL=0 base:MPT_OT_REFILL=0 l2:MPT_OT_REFILL=0 base l2 w4 l2w4 base7:MPT_OT_REFILL=0 base:MPT_OT_REFILL=0 base > $O/s2_var.log 2>&1; cat $O/s2_var.log
SCENES="scene.xml" bash tools/gpu_variants.sh base:MPT_OT_REFILL=0 base > $O/s2_var2.log 2>&1; cat $O/s2_var2.log
timeout -k 10 700 python3 -m pytest tests/test_gpu_adversarial.py -x -q > $O/s2_adv.log 2>&1; tail -15 $O/s2_adv.log
