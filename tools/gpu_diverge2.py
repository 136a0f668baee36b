import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
sc.buildBVH(); bufs = sc.buffers()
W, H = 1920, 1080
u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount())
for spec in os.environ.get("VARIANTS", "MPT_BUDGETS=6,12,24,48").split(";"):
    k, v = spec.split("=", 1); os.environ[k] = v
    ctx = capi.Context(0); ctx.upload_scene(*bufs); ctx.resize(W, H); ctx.set_uniforms(u)
    os.environ.pop(k)
    prim = None
    for depth in (1, 2, 8):
        ctx.clear_sum(); ctx.reset_stats()
        ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=depth, sample_count=int(os.environ.get("SPP","8")), pipeline=2, flags=capi.FLAG_COUNT_WORK)
        s = ctx.stats(); r = s["rays"]
        if depth == 1: prim = (r, s["wave_node_iters"], s["wave_prim_iters"], s["node_visits"], s["prim_tests"])
        br = r - prim[0]
        extra = ""
        if br:
            extra = "| bounce rays only: node slots %.1f (work %.2f) prim slots %.1f (work %.2f)" % (
                64 * (s["wave_node_iters"] - prim[1]) / br, (s["node_visits"] - prim[3]) / br,
                64 * (s["wave_prim_iters"] - prim[2]) / br, (s["prim_tests"] - prim[4]) / br)
        print("%s depth %d: node slots/ray %.2f prim slots/ray %.2f %s" % (spec, depth, 64 * s["wave_node_iters"] / r, 64 * s["wave_prim_iters"] / r, extra), flush=True)
    ctx.close()
