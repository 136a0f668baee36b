#!/bin/bash
O=gpurun_out; mkdir -p $O
U=$PWD/metalpathtracer_amd/lib/libmpt_hip_u.so
MPT_LIB=$U MPT_OT_REFILL=0 timeout -k 10 300 python3 -m pytest tests/test_gpu_ordered.py -x -q -k "bit_exact or closest_hit_matches or in_place" > $O/s4_ord_u.log 2>&1; tail -3 $O/s4_ord_u.log
MPT_LIB=$U timeout -k 10 300 python3 -m pytest tests/test_gpu_ordered.py -x -q -k "bit_exact or in_place" > $O/s4_ord_urf.log 2>&1; tail -3 $O/s4_ord_urf.log
BVH=1 SCENES="bunny20.xml" bash tools/gpu_variants.sh base:MPT_OT_REFILL=0 u:MPT_OT_REFILL=0 "u:MPT_OT_REFILL=0 MPT_OT_BUDGETS=24" "u:MPT_OT_REFILL=0 MPT_OT_BUDGETS=40" "u:MPT_OT_REFILL=0 MPT_OT_BUDGETS=64" "u:MPT_OT_REFILL=0 MPT_OT_BUDGETS=24 MPT_OT_MIN_ACTIVE=0,32" "u:MPT_OT_REFILL=0 MPT_OT_BUDGETS=24 MPT_OT_MIN_ACTIVE=0,16" u "u:MPT_RF_KNOBS=256,32,40,16,8" "u:MPT_RF_KNOBS=256,52,40,16,8" "u:MPT_RF_KNOBS=256,44,56,24,16" base:MPT_OT_REFILL=0 > $O/s4_var.log 2>&1; cat $O/s4_var.log
SCENES="scene.xml" bash tools/gpu_variants.sh base:MPT_OT_REFILL=0 u:MPT_OT_REFILL=0 u > $O/s4_var2.log 2>&1; cat $O/s4_var2.log
