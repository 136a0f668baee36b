"""Serial vs overlapped renders (two lanes) of the headline step (scratch tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", os.environ.get("SCENE", "scene.xml")), sc); assert st == 0
sc.buildBVH()
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
W, H, spp = 1920, 1080, int(os.environ.get("SPP", "256"))
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
pipe = int(os.environ.get("PIPE", "3"))
kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=spp, pipeline=pipe)
for lane in range(2): ctx.render_async(**kw)
ctx.wait()
for mode in ("serial", "async"):
    ctx.reset_stats(); n = 6
    t0 = time.perf_counter()
    for k in range(n):
        (ctx.render if mode == "serial" else ctx.render_async)(sample_begin=k * spp, **kw)
    ctx.wait(); dt = time.perf_counter() - t0
    print("pipe %d %s: %.2f ms per step" % (pipe, mode, dt * 1e3 / n), flush=True)
