import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
sc.buildBVH()
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
W, H = 1920, 1080
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
for pipe in (1, 2):
    for depth in (1, 8):
        row = []
        for spp in (1, 4, 16, 32, 64):
            ts = []
            for rep in range(4):
                ctx.clear_sum(); ctx.reset_stats()
                ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=depth, sample_count=spp, pipeline=pipe)
                ts.append(ctx.stats()["trace_kernel_ms"])
            row.append("%d spp: %.2f ms" % (spp, min(ts[1:])))
        print("pipe", pipe, "depth", depth, " | ".join(row), flush=True)
