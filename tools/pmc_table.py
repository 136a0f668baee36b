"""Prints the DESIGN.md section-4 table rows from the committed counter profiles of a round:  python tools/pmc_table.py r05"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
for name in ("pmc_wavelocal", "pmc_wavelocal_reference_tree", "pmc_wavelocal_cornell", "pmc_ordered_bunny20", "pmc_ordered_config4", "pmc_ordered", "pmc_wavelocal_bunny20"):
    f = os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, name))
    if not os.path.exists(f):
        print(name, "missing"); continue
    p = json.load(open(f)); c = p["counters"]; rays = float(p["rays_per_launch"]); ms = float(p["kernel_ms"])
    clock = c["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e-3)
    valu = c["SQ_INSTS_VALU"] / rays; salu = c["SQ_INSTS_SALU"] / rays
    issue24 = c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * 2.4e9 * ms * 1e-3)
    issue_own = c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * clock * ms * 1e-3)
    wc = c["SQ_WAVE_CYCLES"]
    hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    print("%-30s %7.2f ms  VALU %.1f + SALU %.1f /ray  issue %.3f (own clock %.3f @ %.2f GHz)  wave: issuing %.2f / waitcnt %.2f / stall %.2f  HBM %.0f B/ray (%.2f of 8 TB/s; rd %.1f GB wr %.1f GB)  lane util %.2f  L2 hit %.2f  VMEM rd/ray %.2f  LDS/ray %.2f" % (
        name, ms, valu, salu, issue24, issue_own, clock / 1e9, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc,
        hbm / rays, hbm / (ms * 1e-3) / 8e12, 2.0 * c["FETCH_SIZE"] * 1024 / 1e9, c["WRITE_SIZE"] * 1024 / 1e9,
        c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]), c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
        c["SQ_INSTS_VMEM_RD"] / rays, c["SQ_INSTS_LDS"] / rays))
f = os.path.join(ROOT, "profiles", "%s_mem_ordered_bunny20.json" % tag)
if os.path.exists(f):
    c = json.load(open(f))["counters"]
    cu_cycles = 256.0 * c["GRBM_GUI_ACTIVE"] / 8.0
    print("mem (bunny x20, 64 spp): TD_TD_BUSY %.2f  TA_TA_BUSY %.2f  TD_TC_STALL/TD_BUSY %.2f  tag look-ups per load %.1f  TCP_PENDING_STALL/GATE_EN1 %.2f  L2 read latency %.0f cycles  TCP latency per load %.0f  loads %.0f M  TCC read req %.0f M" % (
        c["TD_TD_BUSY_sum"] / cu_cycles, c["TA_TA_BUSY_sum"] / cu_cycles, c["TD_TC_STALL_sum"] / c["TD_TD_BUSY_sum"],
        c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["TCP_TA_TCP_STATE_READ_sum"], c["TCP_PENDING_STALL_CYCLES_sum"] / c["TCP_GATE_EN1_sum"],
        c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], c["TCP_TCP_LATENCY_sum"] / c["TCP_TA_TCP_STATE_READ_sum"],
        c["TD_LOAD_WAVEFRONT_sum"] / 1e6, c["TCC_READ_sum"] / 1e6))
