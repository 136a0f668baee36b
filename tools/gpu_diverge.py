"""Divergence diagnostics per pipeline/depth (COUNT build): lane utilisation of the node and primitive loops."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = os.environ.get("SCENE", "scene.xml")
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", scene), sc); assert st == 0
sc.buildBVH()
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
W, H = 1920, 1080
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
for depth in (1, 2, 8):
    for pipe in (1, 2):
        ctx.clear_sum(); ctx.reset_stats()
        ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=depth, sample_count=8, pipeline=pipe, flags=capi.FLAG_COUNT_WORK)
        s = ctx.stats()
        r = s["rays"]
        print("depth %d pipe %d: rays %d | per ray: nodes %.2f prims %.2f | wave trips per ray x64: node %.2f prim %.2f | util node %.1f%% prim %.1f%% | leaf phases/ray x64 %.2f"
              % (depth, pipe, r, s["node_visits"]/r, s["prim_tests"]/r, 64*s["wave_node_iters"]/r, 64*s["wave_prim_iters"]/r,
                 100*s["node_visits"]/(64*s["wave_node_iters"]), 100*s["prim_tests"]/(64*s["wave_prim_iters"]), 64*s["wave_leaf_phases"]/r))
