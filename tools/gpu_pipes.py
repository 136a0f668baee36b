"""Compare pipelines: bit-equality of the image and timing (scratch tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = os.environ.get("SCENE", "scene.xml")
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", scene), sc); assert st == 0
sc.buildBVH()
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
W, H = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
spp = int(os.environ.get("SPP", "32"))
pipes = [int(x) for x in os.environ.get("PIPES", "1,2").split(",")]
ref = None
for depth in [int(x) for x in os.environ.get("DEPTHS", "1,2,8,32").split(",")]:
    ref = None
    for pipe in pipes:
        for rep in range(2):
            ctx.clear_sum(); ctx.reset_stats()
            ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=depth, sample_count=spp, pipeline=pipe, flags=int(os.environ.get("FLAGS","0")))
            st = ctx.stats()
        img = ctx.read_sum()
        same = "" if ref is None else ("same" if np.array_equal(img.view(np.uint32), ref.view(np.uint32)) else "DIFFERENT")
        if ref is None: ref = img
        print("depth %2d pipe %d: total_ms %.2f rays %d -> %.1f Mrays/s %s" % (depth, pipe, st["total_ms"], st["rays"], st["rays"] / st["total_ms"] / 1e3, same), flush=True)
