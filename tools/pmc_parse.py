import csv, glob, collections, sys
for d in sys.argv[1:]:
    fs = glob.glob("%s/*/*_counter_collection.csv" % d) + glob.glob("%s/*_counter_collection.csv" % d)
    if not fs: print(d, "no csv"); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        n = r['Kernel_Name']
        if any(k in n for k in ('k_mega', 'k_wavelocal', 'k_step', 'k_stream')):
            agg[(n.split('(')[0][-30:], r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
    last = {}
    for (n, did), v in agg.items(): last[n] = v
    for n, v in last.items():
        print(d, n, {k: ('%.3g' % x) for k, x in sorted(v.items())})
