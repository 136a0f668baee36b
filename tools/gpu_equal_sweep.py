"""One-off evidence sweep: full-size renders through the closest-first and the reference-order pipeline, every float of
the HDR sums compared (the reference-order pipeline is the one checked against the CPU oracle).  CASES / SEED from the env."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(int(os.environ.get("SEED", "11")))
cases = int(os.environ.get("CASES", "48"))
ctx = capi.Context(0)
scenes = {}
total = 0; bad = 0; t0 = time.time()
for case in range(cases):
    name = str(rng.choice(["scene.xml", "glass.xml", "bunny20.xml", "cornell.xml"]))
    mode = int(rng.choice([host.BVH_REFERENCE_SWEEP, host.BVH_BINNED_CENTROID, host.BVH_GPU_LBVH, host.BVH_DEVICE, host.BVH_DEVICE]))
    os.environ["MPT_LBVH_LEAF"] = str(rng.choice(["2", "2", "4", "6"]))     # (device builds: leaf sizes of both kinds of scene)
    key = name
    if key not in scenes:
        sc = host.Scene(); st, log = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", name), sc); assert st == 0, log
        scenes[key] = sc
    sc = scenes[key]
    host.make_ready(ctx, sc, mode)
    if ctx.accel_info()["ordered_ok"] != 1:
        continue
    W, H = [(1920, 1080), (1280, 720), (2560, 1440), (1000, 1000)][int(rng.integers(0, 4))]
    ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
    bsdf = 1 if name == "glass.xml" or rng.random() < 0.3 else 0
    depth = int(rng.choice([4, 8, 16, 32])); spp = int(rng.choice([32, 64, 128])); sb = int(rng.integers(0, 100000))
    seed = (int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32)))
    img = {}
    for pipe in (capi.PIPE_WAVELOCAL, capi.PIPE_ORDERED):
        ctx.clear_sum(); ctx.reset_stats()
        ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_begin=sb, sample_count=spp, seed=seed, pipeline=pipe)
        img[pipe] = ctx.read_sum(); rays = ctx.stats()["rays"]
    same = np.array_equal(img[capi.PIPE_WAVELOCAL].view(np.uint32), img[capi.PIPE_ORDERED].view(np.uint32))
    total += rays; bad += 0 if same else 1
    print("%3d %-12s tree %d %dx%d spp %3d from %6d depth %2d bsdf %d: %s (%.0f M rays)" % (case, name, mode, W, H, spp, sb, depth, bsdf, "bit-identical" if same else "DIFFERS", rays / 1e6), flush=True)
print("%d cases, %.1f G rays, %d mismatches, %.0f s" % (cases, total / 1e9, bad, time.time() - t0))
sys.exit(1 if bad else 0)
