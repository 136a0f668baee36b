"""Timeline of the last device build in a rocprofv3 kernel trace: per kernel name the first start, calls, busy time and the idle
gaps in front of its launches.   usage: python tools/trace_timeline.py <kernel_trace.csv> [first-kernel-name]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_boxes"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
seg = rows[idx[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
prev_end, agg, order = t0, {}, []
for r in seg:
    n = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    if "rocprim" in r["Kernel_Name"] or "hipcub" in r["Kernel_Name"]: n = "cub (scan / sort)"
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = (int(r["Start_Timestamp"]) - prev_end) / 1e3
    prev_end = max(prev_end, int(r["End_Timestamp"]))
    if n not in agg:
        agg[n] = [0, 0.0, 0.0, (int(r["Start_Timestamp"]) - t0) / 1e3]
        order.append(n)
    agg[n][0] += 1; agg[n][1] += d; agg[n][2] += max(gap, 0)
print("%-42s %12s %6s %10s %12s" % ("kernel", "first at us", "calls", "busy us", "idle before"))
for n in order:
    c, d, g, st = agg[n]
    print("%-42s %12.1f %6d %10.1f %12.1f" % (n, st, c, d, g))
print("span %.1f us, busy %.1f us" % ((prev_end - t0) / 1e3, sum(v[1] for v in agg.values())))
