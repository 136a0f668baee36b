"""Effect of the samples-per-pass split on a big scene (config 4: 1M triangles, glass + mirror), 1/8 tile shard."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location("cfg", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_configs.py"))
src = open(spec.origin).read().replace("\nmain()\n", "\n")
ns = {"__file__": spec.origin}
exec(compile(src, spec.origin, "exec"), ns)
tmp = tempfile.mkdtemp()
ns["heightfield"](os.path.join(tmp, "hf.obj"), 501, 1)
xml = os.path.join(tmp, "big.xml")
open(xml, "w").write("""<Scene>
  <Mesh file="%s/hf.obj" position="0,-10,-30" scale="1.0" albedo="0.7,0.7,0.75" emission="0,0,0" materialType="0" emissionPower="0"/>
  <Mesh file="%s/hf.obj" position="0,35,-60" scale="0.6" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="-15,18,-10" radius="9" albedo="0.95,0.95,0.95" emission="0,0,0" materialType="-1" emissionPower="0"/>
  <Sphere position="15,18,-10" radius="9" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="0,60,-20" radius="10" albedo="0,0,0" emission="1,0.9,0.7" materialType="0" emissionPower="5"/>
</Scene>""" % (tmp, tmp))
sc, tb = ns["load"](xml, host.BVH_BINNED_CENTROID)
for spec in os.environ.get("RUNS", "2048:4096,2048:1024,2048:256").split(","):   # spp:pass_spp[:shards]
    f = spec.split(":")
    spp, ps, sh = int(f[0]), f[1], int(f[2]) if len(f) > 2 else 8
    os.environ["MPT_PASS_SPP"] = ps
    ns["measure"]("1M tris, %d spp, pass spp %s, 1/%d shard" % (spp, ps, sh), sc, 1920, 1080, spp, 16, capi.BSDF_SCATTER, None, sh, reps=1)
