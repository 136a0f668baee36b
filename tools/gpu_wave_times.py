import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
sc.buildBVH()
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
W, H = 1920, 1080
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
L = capi.load()
L.mpt_debug_bind(ctx.h)
for rep in range(2):
    ctx.clear_sum(); ctx.reset_stats()
    ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=int(os.environ.get("SPP","64")), pipeline=2)
st = ctx.stats()
n = int(os.environ.get("WAVES", "6144"))
buf = np.zeros((2, n, 8), np.uint64)
L.mpt_debug_wave_times(buf.ctypes.data_as(C.c_void_p), n)
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
late = np.argsort(exh)[-int(n * 0.02):]
early = np.argsort(exh)[:int(n * 0.5)]
def unpack(a): return np.stack([(a >> np.uint64(12 * k)) & np.uint64(0xFFF) for k in range(6)], 1).astype(np.int64)
for name, sel in (("latest 2%", late), ("earliest 50%", early)):
    st_ = unpack(buf[sel, 4]); lf = unpack(buf[sel, 5])[:, :5]
    print(name, "steps after last claim [prim,L0..L4] mean", st_.mean(0).round(1), "left at exhaust [L0..L4] mean", lf.mean(0).round(1),
          "last blk mean %.0f" % buf[sel, 6].astype(np.float64).mean(), "exh-claim mean %.0f us" % (exh[sel] - claim[sel]).mean(), "drain mean %.0f" % (end[sel] - exh[sel]).mean())

tot = reg[:, :5].sum()
names = ["step choice + claim", "ray fetch (primary generation / ring pop)", "closest hit", "shading", "ring push"]
print("shader-clock cycles by region (sum over waves, %% of the total of %.3g):" % tot)
for i, nm in enumerate(names): print("  %-44s %5.1f %%" % (nm, 100 * reg[:, i].sum() / tot))
print("  inside closest hit: box-test loop %.1f %%, leaf (primitive) loop %.1f %% of the total" % (100 * reg[:, 5].sum() / tot, 100 * reg[:, 6].sum() / tot))
