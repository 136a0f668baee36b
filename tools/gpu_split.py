"""One 256-spp render as 1 / 2 / 4 / 8 asynchronous parts on the two render lanes (scratch): wall time and image identity."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", os.environ.get("SCENE", "scene.xml")), sc); assert st == 0
sc.buildBVH(int(os.environ.get("BVH", "0")))
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
W, H = 1920, 1080
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, pipeline=capi.DEFAULT_PIPELINE)
spp = 256
ref = None
for parts in (1, 2, 4, 8, 1, 2, 4):
    best = 1e9
    for rep in range(4):
        ctx.clear_sum(); ctx.reset_stats(); ctx.synchronize()
        t0 = time.perf_counter()
        if parts == 1:
            ctx.render(sample_begin=0, sample_count=spp, **kw)
        else:
            n = spp // parts
            for k in range(parts):
                ctx.render_async(sample_begin=k * n, sample_count=n, **kw)
            ctx.wait()
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    img = ctx.read_sum()
    if ref is None: ref = img.copy()
    same = np.array_equal(ref.view(np.uint32), img.view(np.uint32))
    print("parts %d: %.2f ms wall, %.1f Mrays/s, image %s" % (parts, best, ctx.stats()["rays"] / best / 1e3, "identical" if same else "DIFFERS"), flush=True)
