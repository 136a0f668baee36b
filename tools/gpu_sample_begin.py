"""Does the time of a pass depend on its first sample index?  (1M-triangle scene, 1/8 shard.)"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_configs.py")).read().replace("\nmain()\n", "\n")
ns = {"__file__": __file__}; exec(compile(src, "gpu_configs.py", "exec"), ns)
tmp = tempfile.mkdtemp()
scene = os.environ.get("SCENE", "1m")
if scene == "1m":
    ns["heightfield"](os.path.join(tmp, "hf.obj"), 501, 1)
    xml = os.path.join(tmp, "big.xml")
    open(xml, "w").write("""<Scene>
  <Mesh file="%s/hf.obj" position="0,-10,-30" scale="1.0" albedo="0.7,0.7,0.75" emission="0,0,0" materialType="0" emissionPower="0"/>
  <Mesh file="%s/hf.obj" position="0,35,-60" scale="0.6" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="-15,18,-10" radius="9" albedo="0.95,0.95,0.95" emission="0,0,0" materialType="-1" emissionPower="0"/>
  <Sphere position="15,18,-10" radius="9" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="0,60,-20" radius="10" albedo="0,0,0" emission="1,0.9,0.7" materialType="0" emissionPower="5"/>
</Scene>""" % (tmp, tmp))
else:
    xml = os.path.join(ns["ASSETS"], scene)
sc, _ = ns["load"](xml, host.BVH_BINNED_CENTROID)
W, H = 1920, 1080
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers()); ctx.resize(W, H)
ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
bsdf = capi.BSDF_SCATTER if scene == "1m" else capi.BSDF_LAMBERT
flags = capi.FLAG_COUNT_WORK if os.environ.get("COUNT") else 0
spp = int(os.environ.get("SPP", "256"))
for sb in [0, 0, 1024, 0, 512, 1024, 2048, 4096, 1 << 20, 0]:
    ctx.clear_sum(); ctx.reset_stats()
    ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=16, sample_begin=sb, sample_count=spp, seed=(1, 0),
               shard_rank=0, shard_count=8, flags=flags)
    s = ctx.stats()
    print("sample_begin %8d: %.1f ms  rays %d  rays/path %.3f  node_visits/ray %.2f prim_tests/ray %.2f" % (
        sb, s["total_ms"], s["rays"], s["rays"] / s["paths"], s["node_visits"] / s["rays"], s["prim_tests"] / s["rays"]), flush=True)
