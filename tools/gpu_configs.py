"""Measures the five BASELINE.json configs on ONE MI355X (absolute Mrays/s, HIP-event time of the whole render).

The 8-GPU configs are emulated per GPU: rank 0's tile shard of 8 is rendered (what each of the 8 GPUs does before the
one RCCL reduce), so the figure is the per-GPU rate; the scene, resolution, spp, depth and materials are the config's.
Usage: python tools/gpu_configs.py [quick]      (quick = 1/8 of the samples, for a smoke run)
"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")
CORNELL_CAM = dict(pos=(0.0, 1.0, 3.4), fwd=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), vfov=40.0)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def heightfield(path, n, seed):
    rng = np.random.default_rng(seed)
    xs = np.linspace(-40, 40, n)
    h = rng.uniform(-0.4, 0.4, (n, n)) + 3.0 * np.sin(xs[:, None] * 0.2) * np.cos(xs[None, :] * 0.17)
    with open(path, "w") as f:
        for i in range(n):
            for j in range(n):
                f.write("v %.5f %.5f %.5f\n" % (xs[j], h[i, j], xs[i]))
        for i in range(n - 1):
            for j in range(n - 1):
                a = i * n + j + 1
                f.write("f %d %d %d\nf %d %d %d\n" % (a, a + 1, a + n, a + 1, a + n + 1, a + n))


def load(xml, mode):
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(xml, sc, ASSETS)
    assert st == 0, log
    t0 = time.perf_counter()
    sc.buildBVH(mode)
    return sc, time.perf_counter() - t0


DEVICE = "device"   # not a Scene::BuildMode: mpt_build_and_upload (build -> render on the device)


def measure(name, sc, W, H, spp, depth, bsdf, cam, shard_count, reps=2, device_build=False):
    ctx = capi.Context(0)
    if device_build:
        prims, mats = sc.packed_primitives()
        ctx.build_and_upload(prims, mats)
        t0 = time.perf_counter()
        ctx.build_and_upload(prims, mats)
        print("    (mpt_build_and_upload: %.1f ms wall, own tree of %d nodes)" % ((time.perf_counter() - t0) * 1e3, ctx.accel_info()["nodes"]))
    else:
        ctx.upload_scene(*sc.buffers())
    ctx.resize(W, H)
    ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), cam=cam))
    best = None
    for r in range(reps + 1):
        ctx.clear_sum(); ctx.reset_stats()
        ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_count=spp, seed=(1, 0),
                   shard_rank=0, shard_count=shard_count)
        s = ctx.stats()
        if r and (best is None or s["total_ms"] < best["total_ms"]): best = s
    img = ctx.read_sum()
    ok = bool(np.isfinite(img).all())
    print("%-58s %9.1f ms (trace kernel %.1f) %7.2f Grays/s  %.3f rays/path  %5.2f Gpaths  launches %d finite=%s" % (
        name, best["total_ms"], best["trace_kernel_ms"], best["rays"] / best["total_ms"] / 1e6, best["rays"] / best["paths"], best["paths"] / 1e9,
        best["trace_launches"], ok), flush=True)
    ctx.close()


def main():
    d = 8 if quick else 1
    sc, _ = load(os.path.join(ASSETS, "cornell.xml"), host.BVH_REFERENCE_SWEEP)
    measure("cfg0 cornell 256x256x16 d32 (GPU run of the CPU config)", sc, 256, 256, 16, 32, capi.BSDF_LAMBERT, CORNELL_CAM, 1)
    for dev, tag in ((False, "reference tree"), (True, "device build")):
        measure("     cornell 1920x1080x256 d8, %s" % tag, sc, 1920, 1080, 256 // d, 8, capi.BSDF_LAMBERT, CORNELL_CAM, 1, device_build=dev)
    sc, _ = load(os.path.join(ASSETS, "scene.xml"), host.BVH_REFERENCE_SWEEP)
    for dev, tag in ((False, "reference tree"), (True, "device build")):
        measure("cfg1 scene.xml 1920x1080x256 d8, %s" % tag, sc, 1920, 1080, 256 // d, 8, capi.BSDF_LAMBERT, None, 1, device_build=dev)
        measure("     scene.xml 1920x1080x256 d32, %s" % tag, sc, 1920, 1080, 256 // d, 32, capi.BSDF_LAMBERT, None, 1, device_build=dev)
    trees = ((host.BVH_REFERENCE_SWEEP, "reference tree"), (host.BVH_BINNED_CENTROID, "binned-SAH tree"), (host.BVH_GPU_LBVH, "GPU tree through the host"),
             (DEVICE, "device build"))
    for mode, tag in trees:
        dev = mode == DEVICE
        sc, tb = load(os.path.join(ASSETS, "bunny20.xml"), host.BVH_REFERENCE_SWEEP if dev else mode)
        if not dev:
            print("bunny20 %s: %d prims, %d nodes, build %.3f s" % (tag, sc.getPrimitiveCount(), sc.getBVHNodeCount(), tb))
        measure("cfg2 bunny20 1920x1080x1024 d8, %s" % tag, sc, 1920, 1080, 1024 // d, 8, capi.BSDF_LAMBERT, None, 1, reps=1, device_build=dev)
        measure("cfg3 bunny20 3840x2160x1024 d8, 1/8 tile shard, %s" % tag, sc, 3840, 2160, 1024 // d, 8, capi.BSDF_LAMBERT, None, 8, reps=1, device_build=dev)
    tmp = tempfile.mkdtemp()
    heightfield(os.path.join(tmp, "hf.obj"), 501, 1)
    xml = os.path.join(tmp, "big.xml")
    open(xml, "w").write("""<Scene>
  <Mesh file="%s/hf.obj" position="0,-10,-30" scale="1.0" albedo="0.7,0.7,0.75" emission="0,0,0" materialType="0" emissionPower="0"/>
  <Mesh file="%s/hf.obj" position="0,35,-60" scale="0.6" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="-15,18,-10" radius="9" albedo="0.95,0.95,0.95" emission="0,0,0" materialType="-1" emissionPower="0"/>
  <Sphere position="15,18,-10" radius="9" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="0,60,-20" radius="10" albedo="0,0,0" emission="1,0.9,0.7" materialType="0" emissionPower="5"/>
</Scene>""" % (tmp, tmp))
    for mode, tag in trees:
        dev = mode == DEVICE
        sc, tb = load(xml, host.BVH_BINNED_CENTROID if dev else mode)
        if not dev:
            print("1M-tri heightfields %s: %d prims, %d nodes, build %.3f s" % (tag, sc.getPrimitiveCount(), sc.getBVHNodeCount(), tb))
        measure("cfg4 1M tris glass+mirror 1920x1080x4096 d16, 1/8 tile shard, %s" % tag, sc, 1920, 1080, 4096 // d, 16,
                capi.BSDF_SCATTER, None, 8, reps=1, device_build=dev)


main()
