import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
ctx = capi.Context(0); host.make_ready(ctx, sc, int(os.environ.get("BVH", str(host.BVH_DEVICE))))   # (the tree bench.py renders on)
W, H = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
L = capi.load()
L.mpt_debug_bind(ctx.h)
flags = capi.FLAG_COUNT_WORK if os.environ.get("COUNT") else 0
for rep in range(2):
    if rep: L.mpt_debug_reset()
    ctx.clear_sum(); ctx.reset_stats()
    ctx.render(rng_mode=int(os.environ.get("RNG", str(capi.RNG_PHILOX))), max_depth=int(os.environ.get("DEPTH", "8")), sample_count=int(os.environ.get("SPP","64")), pipeline=2, flags=flags,
               shard_rank=0, shard_count=int(os.environ.get("SHARDS", "1")))   # SHARDS=8: one GPU's tile shard of an 8-GPU render
st = ctx.stats()
n = int(os.environ.get("WAVES", "6144"))
raw = np.zeros(2 * n * 8 + 128, np.uint64)
L.mpt_debug_wave_times(raw.ctypes.data_as(C.c_void_p), n)
buf = raw[:2 * n * 8].reshape(2, n, 8)
lv = raw[2 * n * 8:].reshape(16, 8).astype(np.float64)
reg = buf[1].astype(np.float64)
buf = buf[0]
t0 = buf[:, 0].min()
start = (buf[:, 0] - t0).astype(np.float64) / 100.0
exh = (buf[:, 1] - t0).astype(np.float64) / 100.0
end = (buf[:, 2] - t0).astype(np.float64) / 100.0
print("kernel ms %.2f" % st["trace_kernel_ms"])
pc = lambda a: "p1 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(a, [1, 50, 90, 99, 100]))
print("wave start (us):      ", pc(start))
print("cursor exhausted (us):", pc(exh))
print("wave end (us):        ", pc(end))
print("drain per wave (us):  ", pc(end - exh))

claim = (buf[:, 3] - t0).astype(np.float64) / 100.0
print("last claim (us):      ", pc(claim))
print("exhaust - last claim: ", pc(exh - claim))
LEVELS = int(os.environ.get("LEVELS", "2"))   # MPT_WL_LEVELS of the build (mpt_kernels.h)
late = np.argsort(exh)[-int(n * 0.02):]
early = np.argsort(exh)[:int(n * 0.5)]
def unpack(a): return np.stack([(a >> np.uint64(12 * k)) & np.uint64(0xFFF) for k in range(6)], 1).astype(np.int64)
for name, sel in (("latest 2%", late), ("earliest 50%", early)):
    st_ = unpack(buf[sel, 4])[:, :LEVELS + 1]; lf = unpack(buf[sel, 5])[:, :LEVELS]
    print(name, "steps after last claim [primary, ring 0..] mean", st_.mean(0).round(1), "left at exhaust [ring 0..] mean", lf.mean(0).round(1),
          "last blk mean %.0f" % buf[sel, 6].astype(np.float64).mean(), "exh-claim mean %.0f us" % (exh[sel] - claim[sel]).mean(), "drain mean %.0f" % (end[sel] - exh[sel]).mean())

import json
J = {"what": "k_wavelocal on scene.xml 1920x1080, %s spp, depth 8: diagnostics build (-DMPT_DEBUG_WAVE_TIMES), s_memtime around the regions of every step "
             "summed over all waves; per-step-kind counters with MPT_FLAG_COUNT_WORK (tools/gpu_wave_times.py)" % os.environ.get("SPP", "64"),
     "build": capi.build_id(), "kernel_ms": st["trace_kernel_ms"], "drain_per_wave_us_p50": float(np.percentile(end - exh, 50)),
     "drain_per_wave_us_p99": float(np.percentile(end - exh, 99)), "last_claim_us_p50": float(np.percentile(claim, 50)),
     "wave_end_us_max": float(end.max())}
tot = reg[:, :5].sum()
names = ["step choice + claim", "ray fetch (primary generation / ring pop)", "closest hit", "shading", "ring push"]
J["cycles_by_region_pct"] = {nm: 100 * reg[:, i].sum() / tot for i, nm in enumerate(names)}
J["cycles_by_region_pct"]["closest hit: box-test loop"] = 100 * reg[:, 5].sum() / tot
J["cycles_by_region_pct"]["closest hit: primitive loop"] = 100 * reg[:, 6].sum() / tot
print("shader-clock cycles by region (sum over waves, %% of the total of %.3g):" % tot)
for i, nm in enumerate(names): print("  %-44s %5.1f %%" % (nm, 100 * reg[:, i].sum() / tot))
print("  inside closest hit: box-test loop %.1f %%, leaf (primitive) loop %.1f %% of the total" % (100 * reg[:, 5].sum() / tot, 100 * reg[:, 6].sum() / tot))

if flags:
    print("per step kind (COUNT build): steps, rays/step, done %, box trips/step, box lane utilisation, prim trips/step, prim lane utilisation, share of all box+prim wave trips")
    names = ["primary"] + ["ring %d" % k for k in range(LEVELS)] + ["drain"]
    tot = (lv[:len(names), 1] * 26 + lv[:len(names), 3] * 70).sum()
    for i, nm in enumerate(names):
        st_, bt, bw, pt, pw, rays, dn = lv[i, :7]
        wl = lv[i, 7]
        if st_ == 0: continue
        J.setdefault("per_step_kind", {})[nm] = {"steps": st_, "rays_per_step": rays / st_, "done_pct": 100 * dn / max(1, rays), "box_trips_per_step": bt / st_,
                                                "box_lane_utilisation_pct": 100 * bw / max(1, 64 * bt), "prim_trips_per_step": pt / st_,
                                                "prim_lane_utilisation_pct": 100 * pw / max(1, 64 * pt), "cost_share_pct": 100 * (bt * 26 + pt * 70) / tot,
                                                "box_slots_waiting_with_a_leaf_pct": 100 * wl / max(1, 64 * bt)}
        print("  %-8s steps %9d  rays/step %5.1f  done %5.1f %%  box trips %7.1f util %4.1f %%  prim trips %6.1f util %4.1f %%  cost share %4.1f %%  box slots waiting with a leaf %4.1f %%" % (
            nm, st_, rays / st_, 100 * dn / max(1, rays), bt / st_, 100 * bw / max(1, 64 * bt), pt / st_, 100 * pw / max(1, 64 * pt),
            100 * (bt * 26 + pt * 70) / tot, 100 * wl / max(1, 64 * bt)))

out = os.environ.get("JSON_OUT")
if out:
    json.dump(J, open(out, "w"), indent=1)
