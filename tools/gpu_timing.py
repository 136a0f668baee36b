"""Timing sweep on a GPU box (scratch tool)."""
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
spp = int(os.environ.get("SPP", "64"))
pipes = [int(x) for x in os.environ.get("PIPES", "0,1").split(",")]
slots = [int(x) for x in os.environ.get("SLOTS", "0").split(",")]
for pipe in pipes:
    for sl in (slots if pipe == 0 else [0]):
        for rep in range(2):
            ctx.clear_sum(); ctx.reset_stats()
            ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=spp, pipeline=pipe, slots_per_iter=sl)
            st = ctx.stats()
        print("pipe", pipe, "slots", sl, "1080p x%d: total_ms %.1f trace_ms %.1f launches %d rays %d -> %.1f Mrays/s" % (spp, st["total_ms"], st["trace_kernel_ms"], st["trace_launches"], st["rays"], st["rays"]/st["total_ms"]/1e3), flush=True)
