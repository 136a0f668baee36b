"""Per-GPU step time at N ranks, emulated on one GPU: render only rank r's tile shard (no reduce)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
sc.buildBVH()
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
W, H = 1920, 1080
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
base = None
for n in (1, 2, 4, 8):
    ts = []
    for rep in range(4):
        ctx.clear_sum(); ctx.reset_stats()
        t0 = time.perf_counter()
        ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=256, shard_rank=n - 1, shard_count=n)
        ts.append((time.perf_counter() - t0) * 1e3)
    s = ctx.stats()
    t = min(ts[1:])
    if base is None: base = t
    print("N=%d: wall %.2f ms (kernel %.2f ms) rays %d -> ideal %.2f ms, efficiency %.0f%% (speed-up %.2fx before the reduce)" % (n, t, s["trace_kernel_ms"], s["rays"], base / n, 100 * base / n / t, base / t))
