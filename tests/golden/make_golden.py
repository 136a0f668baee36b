"""Generates the committed golden fixtures in tests/golden/ with the CPU oracle (oracle/mpt_oracle.cpp).

The reference ships no golden vectors; these fixtures freeze the oracle's output — which is
itself pinned to the reference-derived values of SURVEY.md App. C by tests/test_oracle_pins.py — so that
(a) an accidental change of the oracle shows up on CPU, and (b) the GPU box can check the HIP path against
the same numbers without /root/reference.  Inputs are the committed scene files under assets/.

Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CORNELL_CAM = dict(pos=(0.0, 1.0, 3.4), fwd=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), vfov=40.0)

CASES = [
    # name, scene, W, H, camera, kwargs
    ("scene_literal_seed0_f1", "scene.xml", 96, 54, None, dict(rng_mode=ob.RNG_LITERAL, accumulate=0, max_depth=32), dict(random_seed=(0, 0, 0), frame_count=1)),
    ("scene_literal_host_f1", "scene.xml", 96, 54, None, dict(rng_mode=ob.RNG_LITERAL, accumulate=0, max_depth=32), dict(random_seed="host", frame_count=1)),
    ("scene_philox_8spp_d8", "scene.xml", 96, 54, None, dict(rng_mode=ob.RNG_PHILOX, accumulate=1, max_depth=8, sample_count=8, seed=(1, 0)), dict()),
    ("scene_philox_4spp_d32", "scene.xml", 64, 36, None, dict(rng_mode=ob.RNG_PHILOX, accumulate=1, max_depth=32, sample_count=4, seed=(7, 3)), dict()),
    ("cornell_philox_16spp", "cornell.xml", 64, 64, CORNELL_CAM, dict(rng_mode=ob.RNG_PHILOX, accumulate=1, max_depth=32, sample_count=16, seed=(1, 0)), dict()),
    ("glass_scatter_8spp", "glass.xml", 80, 45, None, dict(rng_mode=ob.RNG_PHILOX, bsdf_mode=ob.BSDF_SCATTER, accumulate=1, max_depth=16, sample_count=8, seed=(1, 0)), dict()),
    ("glass_scatter_all_8spp", "glass.xml", 80, 45, None, dict(rng_mode=ob.RNG_PHILOX, bsdf_mode=ob.BSDF_SCATTER_ALL, accumulate=1, max_depth=16, sample_count=8, seed=(1, 0)), dict()),
    ("glass_scatter_all_literal_f1", "glass.xml", 80, 45, None, dict(rng_mode=ob.RNG_LITERAL, bsdf_mode=ob.BSDF_SCATTER_ALL, accumulate=0, max_depth=16), dict(random_seed="host", frame_count=1)),
    ("bunny20_philox_2spp", "bunny20.xml", 64, 36, None, dict(rng_mode=ob.RNG_PHILOX, accumulate=1, max_depth=8, sample_count=2, seed=(1, 0)), dict()),
]


def main():
    manifest = {}
    for name, scene, W, H, cam, rk, uk in CASES:
        sc = ob.OracleScene()
        assert sc.load_xml(os.path.join(ROOT, "assets", scene)) == 0
        sc.build_bvh()
        buf = sc.buffers()
        uk = dict(uk)
        if uk.get("random_seed") == "host":
            uk["random_seed"] = ob.host_seed_sequence(3)
        u = ob.make_uniforms(W, H, sc.prim_count, sc.triangle_count, cam=cam, **uk)
        img, ct = ob.render(u, buf, threads=8, **rk)
        np.save(os.path.join(HERE, name + ".npy"), img)
        manifest[name] = dict(scene=scene, width=W, height=H, camera=cam, render={k: (list(v) if isinstance(v, tuple) else v) for k, v in rk.items()},
                              uniforms={k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in uk.items()},
                              counters=ct, prims=sc.prim_count, nodes=sc.node_count,
                              fnv1a64="%016x" % ob.fnv1a64(img))
        print(name, img.shape, manifest[name]["fnv1a64"])
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
