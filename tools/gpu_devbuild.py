"""Build -> render without the host: wall / device time of mpt_build_and_upload and the render rate of its tree, next to the
host routes (binned SAH + mpt_upload_scene, mpt_build_bvh + mpt_upload_scene).   usage: python tools/gpu_devbuild.py [spp]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64


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


def render(ctx, sc, bsdf, depth, tag):
    W, H = 1920, 1080
    ctx.resize(W, H)
    ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
    best = None
    for k in range(3):
        ctx.clear_sum(); ctx.reset_stats()
        ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_count=spp, seed=(1, 0))
        s = ctx.stats()
        if k and (best is None or s["total_ms"] < best["total_ms"]): best = s
    img = ctx.read_sum()
    print("    %-44s %8.2f ms  %7.2f Grays/s   own nodes %d" % (tag, best["total_ms"], best["rays"] / best["total_ms"] / 1e6, ctx.accel_info()["nodes"]), flush=True)
    return img


def scene(name, xml, bsdf, depth):
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(xml, sc, os.path.join(ROOT, "assets"))
    assert st == 0, log
    print("%s: %d primitives" % (name, sc.getPrimitiveCount()))
    ctx = capi.Context(0)
    # (images of different trees are not compared: the few rays whose answer depends on the visit order see another order)
    for mode, tag in ((host.BVH_BINNED_CENTROID, "host binned SAH + mpt_upload_scene"), (host.BVH_GPU_LBVH, "mpt_build_bvh (SAH) + mpt_upload_scene")):
        t0 = time.perf_counter(); sc.buildBVH(mode); t1 = time.perf_counter(); ctx.upload_scene(*sc.buffers()); t2 = time.perf_counter()
        print("  %-46s build %.1f ms + buffers/upload %.1f ms" % (tag, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
        render(ctx, sc, bsdf, depth, tag)
    prims, mats = sc.packed_primitives()
    for env, tag in (("lbvh", "mpt_build_and_upload (Karras tree)"), ("ploc", "mpt_build_and_upload (PLOC)"), ("sah", "mpt_build_and_upload (SAH, the default)")):
        os.environ["MPT_GPU_BUILD"] = env
        ctx.build_and_upload(prims, mats)
        best, ms = 1e9, 0
        for k in range(3):
            t0 = time.perf_counter(); ms = ctx.build_and_upload(prims, mats); best = min(best, time.perf_counter() - t0)
        print("  %-46s wall %.1f ms (device %.1f ms)" % (tag, best * 1e3, ms))
        render(ctx, sc, bsdf, depth, tag)
    ctx.close()


scene("bunny x20", os.path.join(ROOT, "assets", "bunny20.xml"), capi.BSDF_LAMBERT, 8)
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
scene("1,000,003 primitives", xml, capi.BSDF_SCATTER, 16)
