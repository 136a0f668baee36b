"""Collects the per-dispatch counter values of tools/pmc_round.sh into one JSON and derives the issue-side figures.
usage: python tools/pmc_collect.py <kernel-substring> <out.json> <pmc dir> [<pmc dir> ...]

Normalisation (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
(x4 = shader cycles) summed over all SIMDs (SQ_BUSY_CYCLES: over the SQs); GRBM_GUI_ACTIVE is the sum over the 8 XCDs
of the cycles the kernel was resident, so clock = GRBM_GUI_ACTIVE / 8 / kernel time; a wave64 VALU instruction occupies
its SIMD-32 for 2 cycles; FETCH_SIZE is in KiB and reports half the bytes of wide reads on gfx950 (x2), WRITE_SIZE in
KiB is exact for 16-byte-per-lane stores."""
import collections
import csv
import glob
import json
import sys

kernel, out = sys.argv[1], sys.argv[2]
c = {}
for d in sys.argv[3:]:
    fs = glob.glob("%s/*/*_counter_collection.csv" % d) + glob.glob("%s/*_counter_collection.csv" % d)
    if not fs:
        print(d, "no csv")
        continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if kernel in r["Kernel_Name"]:
            agg.setdefault(r["Dispatch_Id"], collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    if agg:
        last = list(agg.values())[-1]          # the last dispatch of the kernel in that run
        for k, v in last.items():
            c.setdefault(k, v)
json.dump({"kernel": kernel, "counters": c}, open(out, "w"), indent=1)
print(json.dumps(c))
