"""k_ordered on bunny x20: work per ray by bounce generation (MPT_FLAG_COUNT_WORK at depth 1, 2, 3, 8: differences between depths)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "bunny20.xml"), sc); assert st == 0
sc.buildBVH(host.BVH_BINNED_CENTROID)
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
W, H = 1920, 1080
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
prev = None
for d in (1, 2, 3, 8):
    ctx.clear_sum(); ctx.reset_stats()
    ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=d, sample_count=16, pipeline=capi.PIPE_ORDERED, flags=capi.FLAG_COUNT_WORK)
    s = ctx.stats()
    cur = {k: s[k] for k in ("rays", "node_visits", "aabb_hits", "prim_tests", "tree_parked", "wave_node_iters", "wave_prim_iters", "exact_retraces")}
    dlt = cur if prev is None else {k: cur[k] - prev[k] for k in cur}
    r = max(1, dlt["rays"])
    print("depth %d: +%d rays: node visits %.2f, prim tests %.2f, enter the tree %.3f, wave node trips x64 / ray %.1f, wave prim trips x64 / ray %.1f, node-loop lane util %.2f" % (
        d, dlt["rays"], dlt["node_visits"] / r, dlt["prim_tests"] / r, dlt["tree_parked"] / r, 64.0 * dlt["wave_node_iters"] / r, 64.0 * dlt["wave_prim_iters"] / r,
        dlt["node_visits"] / max(1.0, 64.0 * dlt["wave_node_iters"])), flush=True)
    prev = cur
