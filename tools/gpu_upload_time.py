"""Host-side cost of getting a big scene onto the device: ingest, BVH build, mpt_upload_scene (tree conversion)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_configs.py")).read().replace("\nmain()\n", "\n")
ns = {"__file__": __file__}; exec(compile(src, "gpu_configs.py", "exec"), ns)
tmp = tempfile.mkdtemp()
t0 = time.perf_counter(); ns["heightfield"](os.path.join(tmp, "hf.obj"), 501, 1); t_gen = time.perf_counter() - t0
xml = os.path.join(tmp, "big.xml")
open(xml, "w").write('<Scene><Mesh file="%s/hf.obj" position="0,-10,-30" scale="1.0" albedo="0.7,0.7,0.75" emission="0,0,0"/>'
                     '<Mesh file="%s/hf.obj" position="0,35,-60" scale="0.6" albedo="1,1,1" emission="0,0,0" materialType="1.5"/>'
                     '<Sphere position="0,60,-20" radius="10" albedo="0,0,0" emission="1,0.9,0.7" emissionPower="5"/></Scene>' % (tmp, tmp))
sc = host.Scene()
t0 = time.perf_counter(); st, _ = host.SceneLoader.LoadSceneFromXML(xml, sc); t_load = time.perf_counter() - t0
assert st == 0
for mode, name in ((host.BVH_BINNED_CENTROID, "binned"), (host.BVH_REFERENCE_SWEEP, "reference sweep")):
    t0 = time.perf_counter(); sc.buildBVH(mode); t_build = time.perf_counter() - t0
    t0 = time.perf_counter(); bufs = sc.buffers(); t_pack = time.perf_counter() - t0
    ctx = capi.Context(0)
    t0 = time.perf_counter(); ctx.upload_scene(*bufs); ctx.synchronize(); t_up = time.perf_counter() - t0
    print("%d prims, %d nodes, %s builder: OBJ+XML ingest %.2f s, build %.2f s, pack %.2f s, mpt_upload_scene %.2f s" % (
        sc.getPrimitiveCount(), sc.getBVHNodeCount(), name, t_load, t_build, t_pack, t_up), flush=True)
    ctx.close()
