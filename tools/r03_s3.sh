#!/bin/bash
O=gpurun_out; mkdir -p $O
T=$PWD/metalpathtracer_amd/lib/libmpt_hip_times.so
BVH=1 MPT_LIB=$T MPT_OT_REFILL=0 python3 tools/gpu_ot_times.py bunny20.xml 64 > $O/s3_times_old.log 2>&1; cat $O/s3_times_old.log
BVH=1 MPT_LIB=$T python3 tools/gpu_ot_times.py bunny20.xml 64 > $O/s3_times_rf.log 2>&1; cat $O/s3_times_rf.log
BVH=1 MPT_LIB=$T MPT_RF_KNOBS=128,32,48,16,16 python3 tools/gpu_ot_times.py bunny20.xml 64 > $O/s3_times_rf2.log 2>&1; cat $O/s3_times_rf2.log
BVH=1 SCENES="bunny20.xml" bash tools/gpu_variants.sh base:MPT_OT_REFILL=0 "base:MPT_OT_REFILL=0 MPT_OT_CULL_REL=0.0039" "base:MPT_OT_REFILL=0 MPT_OT_CULL_REL=0.0156" "base:MPT_OT_REFILL=0 MPT_OT_CULL_REL=0.031" base "base:MPT_RF_KNOBS=256,56,40,16,8" "base:MPT_RF_KNOBS=256,32,40,16,8" "base:MPT_RF_KNOBS=256,44,56,16,16" "base:MPT_RF_KNOBS=128,44,40,16,8" "base:MPT_RF_KNOBS=64,44,40,24,8" > $O/s3_var.log 2>&1; cat $O/s3_var.log
