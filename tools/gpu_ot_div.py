"""Divergence of the closest-first walk: lane-level work vs wave-level loop trips (MPT_FLAG_COUNT_WORK)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[1:] or ["scene.xml", "bunny20.xml"]:
    sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", name), sc); assert st == 0
    sc.buildBVH()
    ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
    W, H, spp = 1920, 1080, 16
    ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
    for pipe in (2, 3):
        ctx.clear_sum(); ctx.reset_stats()
        ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=spp, pipeline=pipe, flags=capi.FLAG_COUNT_WORK)
        s = ctx.stats()
        r = s["rays"]
        print("%s pipe %d: rays %d; per ray: node visits %.2f prim tests %.2f; wave trips: node %d prim %d leaf phases %d; "
              "node-loop lane util %.2f, prim-loop lane util %.2f; parked %d" % (
                  name, pipe, r, s["node_visits"] / r, s["prim_tests"] / r, s["wave_node_iters"], s["wave_prim_iters"], s["wave_leaf_phases"],
                  s["node_visits"] / max(1, 64 * s["wave_node_iters"]), s["prim_tests"] / max(1, 64 * s["wave_prim_iters"]), s["tree_parked"]), flush=True)
    ctx.close()
