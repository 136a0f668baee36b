#!/bin/bash
O=gpurun_out; mkdir -p $O
for v in r32 r100; do echo "== PLOC radius $v"; MPT_LIB=$PWD/metalpathtracer_amd/lib/libmpt_hip_$v.so timeout -k 10 600 python3 tools/gpu_devbuild.py 64 2>&1 | grep -A1 "PLOC)" ; done > $O/s8_ploc.log 2>&1; cat $O/s8_ploc.log
