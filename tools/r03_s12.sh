#!/bin/bash
O=$PWD/gpurun_out; mkdir -p $O; R=$PWD
timeout -k 10 800 python3 -m pytest tests/test_gpu_lbvh.py -x -q > $O/s12_lbvh.log 2>&1; tail -8 $O/s12_lbvh.log
timeout -k 10 600 python3 tools/gpu_devbuild.py 64 > $O/s12_devbuild.log 2>&1; grep -A1 "build_and_upload\|binned" $O/s12_devbuild.log
cd /tmp && export TMPDIR=/tmp && rm -rf $O/devb_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/devb_trace -o b -- python3 $R/tools/prof_devbuild.py > $O/s12.log 2>&1
cd $R; grep build $O/s12.log
f=$(find $O/devb_trace -name "*kernel_stats.csv" | head -1); cp $f $O/s12_kernel_stats.csv
