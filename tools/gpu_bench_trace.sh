#!/bin/bash
# Kernel timeline of bench.py's timed steps (two renders in flight) for one library build:  tools/gpu_bench_trace.sh <name> [ENV=..]
n=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/bt_$n; rm -rf $OUT; mkdir -p $OUT
lib=$ROOT/metalpathtracer_amd/lib/libmpt_hip_$n.so; [ "$n" = base ] && lib=$ROOT/metalpathtracer_amd/lib/libmpt_hip.so
export MPT_LIB=$lib "$@"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o b -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra-workloads --steps 8 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 1; }
cd $ROOT
python3 - $(find $OUT -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "k_wavelocal" in r["Kernel_Name"]]
t0 = int(ks[0]["Start_Timestamp"])
for r in rows:
    s = (int(r["Start_Timestamp"]) - t0) / 1e6; e = (int(r["End_Timestamp"]) - t0) / 1e6
    if s < 60 or s > 260: continue
    print("%8.2f -> %8.2f  %6.2f ms  queue %s  %s" % (s, e, e - s, r.get("Queue_Id"), r["Kernel_Name"][:60]))
PY
