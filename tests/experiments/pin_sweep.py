"""Which operation convention of MSL's normalize() / dot() reproduces SURVEY.md App. C.3 best?  (test infrastructure)

The survey recorded per-ray work counters of the reference's shader text (1280x720, host seed, literal RNG).  The oracle
misses them by 2 rays of 1.5 M.  This script rebuilds the oracle with the alternative conventions it can select
(-DORC_NORMALIZE=0..3, -DORC_DOT=0..2) and prints the residuals.  Result (round 2): the default (v * (1/len), left-to-right
dot) is the closest by far — rays -2, node pops +24, misses and emissive hits exact; v/len gives rays -6 / pops -582, the
rsqrt forms -11..-29 / -587..-1467 — and none is exact, so the pin stays at "within 5e-5" (tests/test_oracle_pins.py).

  python tests/experiments/pin_sweep.py
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WANT = dict(rays=1555067, node_pops=13145685, aabb_pass=8490497, prim_tests=6152603, sphere_tests=3560094,
            tri_tests=2592509, bounces=634744, misses=920323, emissive_hits=74701)
CHILD = r'''
import sys
sys.path.insert(0, %r)
from oracle import binding as ob
ob._LIB_PATH = sys.argv[1]
ob.build = lambda force=False: sys.argv[1]
sc = ob.OracleScene(); assert sc.load_xml(%r) == 0; sc.build_bvh(); buf = sc.buffers()
u = ob.make_uniforms(1280, 720, sc.prim_count, sc.triangle_count, random_seed=ob.host_seed_sequence(3), frame_count=1)
_, ct = ob.render(u, buf, rng_mode=ob.RNG_LITERAL, accumulate=0, threads=8)
print({k: ct[k] - v for k, v in %r.items()})
''' % (ROOT, os.path.join(ROOT, "assets", "scene.xml"), WANT)

with tempfile.TemporaryDirectory() as tmp:
    for nrm in range(4):
        for dot in range(3):
            so = os.path.join(tmp, "orc_%d_%d.so" % (nrm, dot))
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                                   "-DORC_NORMALIZE=%d" % nrm, "-DORC_DOT=%d" % dot, "-shared", "-o", so,
                                   os.path.join(ROOT, "oracle", "mpt_oracle.cpp"), "-lpthread"])
            out = subprocess.check_output([sys.executable, "-c", CHILD, so], text=True)
            print("normalize %d dot %d: residual vs App. C.3 %s" % (nrm, dot, out.strip()), flush=True)
