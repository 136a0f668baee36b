#!/bin/bash
# bench.py's headline step (overlapped renders) for several library builds / knobs on one box:  tools/gpu_bench_ab.sh "name[:ENV=..]" ...
for v in "$@"; do
  n=${v%%:*}; e=${v#*:}; [ "$e" = "$v" ] && e=""
  lib=$PWD/metalpathtracer_amd/lib/libmpt_hip_$n.so; [ "$n" = base ] && lib=$PWD/metalpathtracer_amd/lib/libmpt_hip.so
  env MPT_LIB=$lib $e python3 bench.py --no-cpu-baseline --no-extra-workloads --steps 8 --warmup 2 2>/tmp/bab_$$.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d.get('roofline') or {}
print('$n [$e]: %.0f Mrays/s  %.2f ms/step  serial %.2f ms  avg launch %.2f ms' % (d['value'], d['ms_per_step'], d.get('serial_ms_per_render') or -1, r.get('avg_launch_ms') or -1))" || tail -3 /tmp/bab_$$.err
done
rm -f /tmp/bab_$$.err
