"""Would overlapping consecutive renders pay?  Two contexts on one GPU, one host thread each, vs one context."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
sc.buildBVH(); bufs = sc.buffers()
W, H = 1920, 1080
u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount())
def mk():
    c = capi.Context(0); c.upload_scene(*bufs); c.resize(W, H); c.set_uniforms(u); return c
ctxs = [mk() for _ in range(int(os.environ.get("NCTX", "2")))]
def loop(c, n, spp, shards):
    for i in range(n):
        c.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_begin=i * spp, sample_count=spp, shard_rank=0, shard_count=shards)
for spp, shards, n in ((256, 1, 12), (256, 8, 24)):
    for c in ctxs: loop(c, 1, spp, shards)
    t0 = time.perf_counter(); loop(ctxs[0], n, spp, shards); t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(c, n // len(ctxs), spp, shards)) for c in ctxs]
    [t.start() for t in th]; [t.join() for t in th]
    t2 = time.perf_counter() - t0
    print("spp %d shard 1/%d: %d renders, one context %.2f ms/render, %d overlapped contexts %.2f ms/render (%.1f%%)" % (
        spp, shards, n, t1 * 1e3 / n, len(ctxs), t2 * 1e3 / n, 100 * (t1 / t2 - 1)), flush=True)
