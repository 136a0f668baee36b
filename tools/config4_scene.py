"""BASELINE.json configs[4]'s scene, written as XML + OBJ in the reference's schema: two 500 K-triangle height fields (one of
them glass), a mirror and a glass sphere, an emitter — 1,000,003 primitives.  Used by bench.py (extra workload), tools/prof_run.py
(SCENE=config4) and tools/gpu_configs.py; the GPU tests write the same files (tests/test_gpu_parity.py _heightfield_obj)."""
import os

import numpy as np


def heightfield_obj(path, n=501, seed=1):
    rng = np.random.default_rng(seed)
    xs = np.linspace(-40, 40, n)
    h = rng.uniform(-0.4, 0.4, (n, n)) + 3.0 * np.sin(xs[:, None] * 0.2) * np.cos(xs[None, :] * 0.17)
    v = np.stack([np.broadcast_to(xs[None, :], (n, n)), h, np.broadcast_to(xs[:, None], (n, n))], -1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j + 1).reshape(-1)
    f = np.stack([np.stack([a, a + 1, a + n], -1), np.stack([a + 1, a + n + 1, a + n], -1)], 1).reshape(-1, 3)
    with open(path, "w") as out:
        np.savetxt(out, v, fmt="v %.5f %.5f %.5f")
        np.savetxt(out, f, fmt="f %d %d %d")


def write(dirpath):
    """Writes hf.obj + config4.xml into dirpath and returns the XML path."""
    os.makedirs(dirpath, exist_ok=True)
    heightfield_obj(os.path.join(dirpath, "hf.obj"))
    xml = os.path.join(dirpath, "config4.xml")
    open(xml, "w").write("""<Scene>
  <Mesh file="hf.obj" position="0,-10,-30" scale="1.0" albedo="0.7,0.7,0.75" emission="0,0,0" materialType="0" emissionPower="0"/>
  <Mesh file="hf.obj" position="0,35,-60" scale="0.6" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="-15,18,-10" radius="9" albedo="0.95,0.95,0.95" emission="0,0,0" materialType="-1" emissionPower="0"/>
  <Sphere position="15,18,-10" radius="9" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="0,60,-20" radius="10" albedo="0,0,0" emission="1,0.9,0.7" materialType="0" emissionPower="5"/>
</Scene>""")
    return xml
