"""Serial step time of every rank's tile shard at N = 2, 4, 8 (one GPU emulating each rank in turn)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
sc.buildBVH()
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
W, H = 1920, 1080
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
for n in (1, 2, 4, 8):
    out = []
    for r in range(n):
        kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=256, shard_rank=r, shard_count=n)
        best = 1e9
        for rep in range(3):
            ctx.reset_stats(); ctx.render(**kw); s = ctx.stats(); best = min(best, s["trace_kernel_ms"])
        out.append("%.2f(%.0fM)" % (best, s["rays"] / 1e6))
    print("N=%d kernel ms per rank (rays):" % n, " ".join(out), flush=True)
