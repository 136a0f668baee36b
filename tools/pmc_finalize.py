"""profiles/<out>.json from one tools/pmc_round.sh run: counters of the last dispatch of the kernel + the run's own rays per
launch and kernel time (HIP events of tools/prof_run.py in the first counter pass, the pass that also holds GRBM_GUI_ACTIVE)
+ the derived figures bench.py uses.   usage: python tools/pmc_finalize.py <tag> <out.json> [note]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
pmc = json.load(open(os.path.join(ROOT, "gpurun_out", tag + "_pmc.json")))
log = open(os.path.join(ROOT, "gpurun_out", "pmc_%s_1.log" % tag)).read()
m = re.findall(r"pipe (\d+) spp (\d+) depth (\d+): total_ms ([\d.]+) trace_ms ([\d.]+) launches (\d+) rays (\d+)", log)[-1]
wl = re.findall(r"^WORKLOAD (\{.*\})$", log, flags=re.M)
workload = json.loads(wl[-1]) if wl else None
c = pmc["counters"]
rays, ms = int(m[6]) / int(m[5]), float(m[4]) / int(m[5])
clock = c["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e-3) / 1e9
d = {
    "note": ("rocprofv3 --pmc, one counter set per pass, no tracing (tools/pmc_round.sh on tools/prof_run.py); counters are of ONE launch of "
             "%s (the last of the pass); kernel_ms = HIP-event time of that launch in the pass that collected GRBM_GUI_ACTIVE. %s" % (pmc["kernel"], note)).strip(),
    "kernel": pmc["kernel"], "pipeline": int(m[0]), "spp": int(m[1]), "depth": int(m[2]),
    "workload": workload,   # scene / size / spp / depth / pipeline / tree builder and the build (library + source sha256): bench.py
                            # uses a profile only for a run of the same workload with the same build
    "rays_per_launch": rays, "kernel_ms": ms, "counters": c,
    "derived": {
        "clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / kernel time": clock,
        "valu_wave_instr_per_ray": c["SQ_INSTS_VALU"] / rays,
        "salu_per_valu": c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"],
        "valu_lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU)": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]),
        "valu_issue_frac = SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)": c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0),
        "wave_cycles: issuing / s_waitcnt / issue stall (SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES)": [
            c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]],
        "lds_busy_frac = SQ_LDS_IDX_ACTIVE / 256 CUs / (GRBM_GUI_ACTIVE / 8)": c["SQ_LDS_IDX_ACTIVE"] / 256.0 / (c["GRBM_GUI_ACTIVE"] / 8.0),
        "lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
        "hbm_read_bytes = 2 * FETCH_SIZE KiB (gfx950 correction)": 2.0 * c["FETCH_SIZE"] * 1024.0,
        "hbm_write_bytes = WRITE_SIZE KiB": c["WRITE_SIZE"] * 1024.0,
        "hbm_bytes_per_ray": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / rays,
        "hbm_frac_of_8TBs": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / (ms * 1e-3) / 8e12,
        "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    },
}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d["derived"], indent=1))
