import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from metalpathtracer_amd import capi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = ob.OracleScene(); sc.load_xml(os.path.join(ROOT, "assets/scene.xml")); sc.build_bvh(); buf = sc.buffers()
ctx = capi.Context(0)
ctx.upload_scene(*buf)
W, H = 1920, 1080
uo = ob.make_uniforms(W, H, sc.prim_count, sc.triangle_count)
ctx.resize(W, H); ctx.set_uniforms(capi.Uniforms.from_buffer_copy(bytes(uo)))
spp = int(os.environ.get("SPP", "32"))
for depth in (1, 2, 3, 8, 32):
    for pipe in (0, 1):
        for rep in range(2):
            ctx.clear_sum(); ctx.reset_stats()
            ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=depth, sample_count=spp, pipeline=pipe, slots_per_iter=int(os.environ.get("SLOTS","0")))
            st = ctx.stats()
        print("depth", depth, "pipe", pipe, "total_ms %.2f launches %d rays %d -> %.1f Mrays/s" % (st["total_ms"], st["trace_launches"], st["rays"], st["rays"]/st["total_ms"]/1e3), flush=True)
