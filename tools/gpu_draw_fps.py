"""Frame protocol speed: mpt_draw (1 sample/pixel per frame, running mean — what the reference's Renderer::draw does)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
sc.buildBVH()
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
for W, H in ((1280, 720), (1920, 1080), (3840, 2160)):
    ctx.resize(W, H)
    seeds = host.host_seed_sequence(3)
    n = 200
    for rng_mode, name in ((capi.RNG_LITERAL, "literal"), (capi.RNG_PHILOX, "philox")):
        for f in range(n + 10):
            if f == 10: ctx.synchronize(); t0 = time.perf_counter()
            ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), random_seed=seeds, frame_count=f))
            ctx.draw(rng_mode=rng_mode, max_depth=32, sample_begin=f, sample_count=1, pipeline=int(os.environ.get("PIPE", "2")))
        ctx.synchronize(); dt = time.perf_counter() - t0
        print("%dx%d %s: %.3f ms/frame (%.0f frames/s, %.1f Mpaths/s)" % (W, H, name, dt * 1e3 / n, n / dt, W * H * n / dt / 1e6), flush=True)
