#!/bin/bash
O=$PWD/gpurun_out; mkdir -p $O
PIPE=2 SCENES="scene.xml" bash tools/gpu_variants.sh base split masked maskedsplit base split masked maskedsplit > $O/s13_var.log 2>&1; cat $O/s13_var.log
BVH=1 PIPE=3 SCENES="bunny20.xml" bash tools/gpu_variants.sh base split base split > $O/s13_var2.log 2>&1; cat $O/s13_var2.log
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $O/s13_gpu.log 2>&1; tail -15 $O/s13_gpu.log
