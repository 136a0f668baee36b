"""Per-GPU step time at N ranks, emulated on one GPU: render only rank r's tile shard (no reduce)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
ctx = capi.Context(0); host.make_ready(ctx, sc, int(os.environ.get("BVH", str(host.BVH_DEVICE))))   # (the tree bench.py renders on)
W, H = 1920, 1080
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
base = None
K = 6
for n in (1, 2, 4, 8):
    kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=256, shard_rank=n - 1, shard_count=n)
    ctx.render_async(**kw); ctx.render_async(**kw); ctx.wait()           # both lanes allocated and warm
    ts, ta = [], []
    for rep in range(3):
        ctx.clear_sum(); ctx.reset_stats(); ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(K): ctx.render(sample_begin=256 * k, **kw)         # serial steps
        ts.append((time.perf_counter() - t0) * 1e3 / K)
        ctx.clear_sum(); ctx.reset_stats(); ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(K): ctx.render_async(sample_begin=256 * k, **kw)   # overlapped steps (what bench.py does)
        ctx.wait()
        ta.append((time.perf_counter() - t0) * 1e3 / K)
    s = ctx.stats()
    t, a = min(ts), min(ta)
    if base is None: base = (t, a)
    print("N=%d: per step serial %.2f ms (%.2fx), overlapped %.2f ms (%.2fx of the overlapped 1-GPU step) — before the reduce; kernel avg %.2f ms" % (
        n, t, base[0] / t, a, base[1] / a, s["trace_kernel_ms"] / max(1, s["trace_launches"])))
