"""Own 4-wide tree built by the host (binned SAH over the leaves, mpt_upload_scene) vs collapsed on the device
(mpt_build_and_upload) over the SAME binary tree: node visits / primitive tests per ray and render time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "bunny20.xml"), sc); assert st == 0
sc.buildBVH(host.BVH_BINNED_CENTROID)
prims, mats = sc.packed_primitives()
ctx = capi.Context(0)
W, H = 1920, 1080
def run(tag):
    ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
    ctx.clear_sum(); ctx.reset_stats()
    ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=8, seed=(1, 0), pipeline=capi.PIPE_ORDERED, flags=capi.FLAG_COUNT_WORK)
    s = ctx.stats(); i = ctx.accel_info()
    ctx.clear_sum(); ctx.reset_stats()
    ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=64, seed=(1, 0), pipeline=capi.PIPE_ORDERED)
    t = ctx.stats()
    print("%-40s own nodes %6d depth %2d lds nodes %3d | per ray: node visits %.3f box hits %.3f prim tests %.3f | retraced %.2e parked %.3f | 64 spp %.2f ms" % (
        tag, i["nodes"], i["depth"], i["lds_nodes"], s["node_visits"] / s["rays"], s["aabb_hits"] / s["rays"], s["prim_tests"] / s["rays"],
        s["exact_retraces"] / s["rays"], s["tree_parked"] / s["rays"], t["total_ms"]), flush=True)
ctx.upload_scene(*sc.buffers()); run("host binned tree, host own tree")
for env in ("lbvh", "ploc"):
    os.environ["MPT_GPU_BUILD"] = env
    ctx.build_and_upload(prims, mats); run("device %s, device collapse" % env)
    bvh, idx = ctx.download_bvh()
    ctx.upload_scene(bvh, prims, mats, idx); run("device %s tree, host own tree" % env)
