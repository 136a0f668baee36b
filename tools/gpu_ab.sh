#!/bin/bash
# A/B two library builds on the SAME box, alternating processes.  usage: gpu_ab.sh libA libB [SPP]
A=$1; B=$2; export SPP=${3:-256}
for i in 1 2 3; do
  for L in $A $B; do
    echo -n "$(basename $L): "; MPT_LIB=$PWD/$L REPS=3 PIPE=2 python tools/prof_run.py | tail -1
  done
done
