"""Tree builders x pipelines on the big configs (BASELINE.json configs 2 and 4 scenes, reduced spp): build time, upload
time (tree conversion + own-tree build + copy), Grays/s.  usage: python tools/gpu_trees.py [spp]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
ns = {"__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_configs.py")}
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_configs.py")).read().split("def load(")[0]
exec(src, ns)
ROOT, ASSETS = ns["ROOT"], ns["ASSETS"]
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tmp = tempfile.mkdtemp()
ns["heightfield"](os.path.join(tmp, "hf.obj"), 501, 1)
big = os.path.join(tmp, "big.xml")
open(big, "w").write("""<Scene>
  <Mesh file="%s/hf.obj" position="0,-10,-30" scale="1.0" albedo="0.7,0.7,0.75" emission="0,0,0" materialType="0" emissionPower="0"/>
  <Mesh file="%s/hf.obj" position="0,35,-60" scale="0.6" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="-15,18,-10" radius="9" albedo="0.95,0.95,0.95" emission="0,0,0" materialType="-1" emissionPower="0"/>
  <Sphere position="15,18,-10" radius="9" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="0,60,-20" radius="10" albedo="0,0,0" emission="1,0.9,0.7" materialType="0" emissionPower="5"/>
</Scene>""" % (tmp, tmp))
ctx = capi.Context(0)
for label, xml, depth, bsdf, shards in (("bunny x20 (99,362 prims) 1920x1080 d8", os.path.join(ASSETS, "bunny20.xml"), 8, 0, 1),
                                        ("1 M triangles glass+mirror 1920x1080 d16, 1/8 tile shard", big, 16, 1, 8)):
    for tree, mode, env in (("reference sweep", host.BVH_REFERENCE_SWEEP, None), ("binned SAH (host)", host.BVH_BINNED_CENTROID, None),
                            ("GPU LBVH leaf<=8", host.BVH_GPU_LBVH, "8"), ("GPU LBVH leaf<=4", host.BVH_GPU_LBVH, "4"),
                            ("GPU LBVH leaf<=2", host.BVH_GPU_LBVH, "2")):
        if env: os.environ["MPT_LBVH_LEAF"] = env
        sc = host.Scene()
        st, log = host.SceneLoader.LoadSceneFromXML(xml, sc, ASSETS); assert st == 0, log
        t0 = time.perf_counter(); sc.buildBVH(mode); tb = time.perf_counter() - t0
        buf = sc.buffers()
        t0 = time.perf_counter(); ctx.upload_scene(*buf); tu = time.perf_counter() - t0
        info = ctx.accel_info()
        W, H = 1920, 1080
        ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
        res = []
        for pipe in (capi.PIPE_WAVELOCAL, capi.PIPE_ORDERED):
            best = 1e9
            for rep in range(2):
                ctx.clear_sum(); ctx.reset_stats()
                ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_count=spp, pipeline=pipe, shard_rank=0, shard_count=shards)
                s = ctx.stats(); best = min(best, s["total_ms"])
            res.append("%s %.1f ms %.2f Grays/s" % ("reference-order" if pipe == 2 else "closest-first", best, s["rays"] / best / 1e6))
        print("%s | %-18s build %.3f s, %d ref nodes, upload %.3f s (own tree %d nodes, depth %d) | %s | %s" % (
            label, tree, tb, sc.getBVHNodeCount(), tu, info["nodes"], info["depth"], res[0], res[1]), flush=True)
