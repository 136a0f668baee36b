"""Randomised parity sweep: HIP path vs the CPU oracle, bit for bit, over scenes / sizes / sample ranges / depths /
seeds / shard counts / pipelines (run by tests/test_gpu_parity.py::test_randomised_cases_bit_exact; CASES / SEED from
the environment for longer one-off sweeps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
from oracle import binding as ob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNELL_CAM = dict(pos=(0.0, 1.0, 3.4), fwd=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), vfov=40.0)
rng = np.random.default_rng(int(os.environ.get("SEED", "7")))
scenes = {}
for name in ("scene.xml", "cornell.xml", "glass.xml", "bunny20.xml"):
    sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", name), sc); assert st == 0
    sc.buildBVH(); scenes[name] = (sc, sc.buffers())
ctx = capi.Context(0)
n_cases = int(os.environ.get("CASES", "40")); bad = 0; t0 = time.time()
for case in range(n_cases):
    name = str(rng.choice(list(scenes)))
    sc, buf = scenes[name]
    W, H = int(rng.integers(9, 260)), int(rng.integers(9, 150))
    if name == "bunny20.xml": W, H = min(W, 96), min(H, 64)
    spp = int(rng.integers(1, 10)); sb = int(rng.integers(0, 5000)); depth = int(rng.choice([1, 2, 3, 8, 16, 32]))
    seed = (int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32)))
    bsdf = int(rng.choice([1, 1, 2])) if name == "glass.xml" or rng.random() < 0.2 else 0
    pipe = int(rng.choice([3, 3, 3, 3, 2, 2, 0, 1])); shards = int(rng.choice([1, 1, 2, 3, 5]))
    cam = CORNELL_CAM if name == "cornell.xml" else None
    ctx.upload_scene(*buf); ctx.resize(W, H)
    u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), cam=cam)
    ctx.set_uniforms(u); ctx.clear_sum()
    for r in range(shards):
        fn = ctx.render_async if (pipe >= 2 and rng.random() < 0.5) else ctx.render
        fn(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_begin=sb, sample_count=spp, seed=seed,
           pipeline=pipe, shard_rank=r, shard_count=shards)
    got = ctx.read_sum()
    ref, _ = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buf, rng_mode=ob.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth,
                       accumulate=1, sample_begin=sb, sample_count=spp, seed=seed, threads=16)
    ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    bad += not ok
    print("%2d %-12s %3dx%-3d spp %d from %4d depth %2d bsdf %d pipe %d shards %d : %s" % (
        case, name, W, H, spp, sb, depth, bsdf, pipe, shards, "bit-identical" if ok else "MISMATCH (%d floats)" % int((got != ref).sum())), flush=True)
print("%d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
