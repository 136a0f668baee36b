"""Work counters per ray for the tree builders on bunny x20 (MPT_FLAG_COUNT_WORK, 4 spp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ctx = capi.Context(0)
for tree, mode, env in (("reference sweep", 0, None), ("binned", 1, None), ("LBVH<=8", 2, "8"), ("LBVH<=4", 2, "4")):
    if env: os.environ["MPT_LBVH_LEAF"] = env
    sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "bunny20.xml"), sc); assert st == 0
    sc.buildBVH(mode)
    bvh, prims, mats, idx = [np.asarray(a) for a in sc.buffers()]
    bvh = bvh.reshape(-1, 8); cnt = bvh[:, 7].copy().view(np.int32); leaves = cnt > 0
    ext = bvh[leaves, 4:7] - bvh[leaves, 0:3]
    area = 2 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0])
    big = np.sort(area)[-5:]
    print("%s: %d nodes, %d leaves, prims/leaf %.2f, leaf area median %.3f mean %.3f, largest %s" % (tree, bvh.shape[0], leaves.sum(), cnt[leaves].mean(), np.median(area), area.mean(), np.array2string(big, precision=1)))
    ctx.upload_scene(*sc.buffers()); ctx.resize(960, 540); ctx.set_uniforms(host.make_uniforms(960, 540, sc.getPrimitiveCount(), sc.getTriangleCount()))
    for pipe in (2, 3):
        ctx.clear_sum(); ctx.reset_stats()
        ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=4, pipeline=pipe, flags=capi.FLAG_COUNT_WORK)
        s = ctx.stats(); r = s["rays"]
        print("   pipe %d: node visits/ray %.2f, box hits/ray %.2f, prim tests/ray %.2f, retraced %d" % (pipe, s["node_visits"] / r, s["aabb_hits"] / r, s["prim_tests"] / r, s["exact_retraces"]), flush=True)
