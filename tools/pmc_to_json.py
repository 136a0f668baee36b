"""profiles/<tag>_pmc_hbm_traffic.json from the raw counter sums of tools/profile_round.sh.
usage: python tools/pmc_to_json.py gpurun_out/r01b_pmc_raw.txt gpurun_out/r01b_kernel_stats.csv gpurun_out/r01b_bench_under_rocprof.json profiles/r01_pmc_hbm_traffic.json"""
import ast, csv, json, sys
raw, stats, bench, out = sys.argv[1:5]
c = {}
for line in open(raw):
    if "k_wavelocal" in line:
        c.update({k: float(v) for k, v in ast.literal_eval(line[line.index("{"):]).items()})
row = next(r for r in csv.DictReader(open(stats)) if "k_wavelocal" in r["Name"])
b = json.loads([l for l in open(bench) if l.startswith("{")][-1])
rays = b["config"]["rays"] // b["steps"]
rd, wr = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
j = {
    "note": "rocprofv3 --pmc passes (one counter set per run, no tracing; tools/profile_round.sh) on tools/prof_run.py: scene.xml "
            "1920x1080, 256 spp, depth 8, philox, wave-local pipeline; values are for ONE k_wavelocal launch.  Kernel time: "
            "rocprofv3 --kernel-trace --stats of `python3 bench.py` gives avg %.2f ms over %s launches (kernel_stats.csv); the "
            "bench line printed in that run derives %.2f ms from its own HIP events." % (
                float(row["AverageNs"]) / 1e6, row["Calls"],
                b["roofline"]["avg_launch_ms"]),
    "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
    "TCC_HIT_sum": c["TCC_HIT_sum"], "TCC_MISS_sum": c["TCC_MISS_sum"],
    "SQ": {k: v for k, v in c.items() if k.startswith("SQ_")},
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read stream (MI355X_MICROARCH.md, HBM) -> read "
                  "bytes = 2 * FETCH_SIZE * 1024 (upper estimate; our reads are 16 B/lane ring pops and scattered 16 B primitive "
                  "fetches); WRITE_SIZE is exact for 16-B-per-lane stores",
    "rays_per_launch": rays,
    "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
    "traffic_bytes_per_ray": (rd + wr) / rays,
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "valu_wave_instr_per_ray": c["SQ_INSTS_VALU"] / rays,
    "valu_lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]),
}
json.dump(j, open(out, "w"), indent=1)
print(json.dumps({k: j[k] for k in ("traffic_bytes_per_ray", "l2_hit_rate", "valu_wave_instr_per_ray", "valu_lane_utilisation")}))
