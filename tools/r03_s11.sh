#!/bin/bash
O=$PWD/gpurun_out; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf $O/devb_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/devb_trace -o b -- python3 $R/tools/prof_devbuild.py > $O/s11.log 2>&1
cd $R; tail -5 $O/s11.log
f=$(find $O/devb_trace -name "*kernel_stats.csv" | head -1); cp $f $O/s11_kernel_stats.csv; head -30 $O/s11_kernel_stats.csv | cut -c1-150
