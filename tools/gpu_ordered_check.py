"""First-light check of MPT_PIPE_ORDERED on a GPU box: images vs the reference-order pipeline (bit for bit), closest
hits of random rays vs mpt_trace_rays, then timing at 1080p (scratch tool; the pytest suite holds the real checks)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", name), sc)
    assert st == 0, log
    sc.buildBVH(int(os.environ.get("BVH", "0")))
    return sc


def images(ctx, sc, W, H, spp, depth, bsdf):
    ctx.resize(W, H)
    ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
    out = {}
    for pipe in (capi.PIPE_WAVELOCAL, capi.PIPE_ORDERED):
        ctx.clear_sum(); ctx.reset_stats()
        ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_count=spp, pipeline=pipe, flags=capi.FLAG_COUNT_WORK)
        out[pipe] = (ctx.read_sum(), ctx.stats())
    a, sa = out[capi.PIPE_WAVELOCAL]; b, sb = out[capi.PIPE_ORDERED]
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    print("  %dx%d x%d d%d bsdf%d: images %s; rays %d vs %d; ordered: retraced %d parked %d node visits/ray %.2f prim tests/ray %.2f"
          % (W, H, spp, depth, bsdf, "IDENTICAL" if same else "DIFFER (%d floats)" % int((a.view(np.uint32) != b.view(np.uint32)).sum()),
             sa["rays"], sb["rays"], sb["exact_retraces"], sb["tree_parked"], sb["node_visits"] / max(1, sb["rays"]), sb["prim_tests"] / max(1, sb["rays"])), flush=True)
    return same


def rays(ctx, n, seed):
    rng = np.random.default_rng(seed)
    o = np.empty((n, 3), np.float32); d = np.empty((n, 3), np.float32)
    o[:] = rng.normal(size=(n, 3)) * [20, 10, 20] + [0, 12, 10]
    tgt = rng.normal(size=(n, 3)) * [8, 8, 8] + [0, 6, 0]
    d[:] = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: n // 64, 0] = 0.0        # degenerate directions
    t0, p0, n0, f0 = ctx.trace_rays(o, d)
    t1, p1, n1, f1, fl = ctx.trace_rays_ordered(o, d)
    ok = np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and np.array_equal(p0, p1) and np.array_equal(n0.view(np.uint32), n1.view(np.uint32)) and np.array_equal(f0, f1)
    print("  %d random rays: %s; hits %d; flags: dir %d tie %d check %d overflow %d" % (n, "IDENTICAL" if ok else "DIFFER (%d)" % int(((t0.view(np.uint32) != t1.view(np.uint32)) | (p0 != p1)).sum()),
          int((p0 >= 0).sum()), int((fl & 1 != 0).sum()), int((fl & 2 != 0).sum()), int((fl & 4 != 0).sum()), int((fl & 8 != 0).sum())), flush=True)
    return ok


def main():
    ctx = capi.Context(0)
    ok = True
    for name, bsdf in (("scene.xml", 0), ("glass.xml", 1), ("cornell.xml", 0), ("bunny20.xml", 0)):
        sc = load(name)
        ctx.upload_scene(*sc.buffers())
        print(name, ctx.accel_info(), flush=True)
        ok &= rays(ctx, 1 << 18, 1)
        ok &= images(ctx, sc, 128, 72, 4, 8, bsdf)
        ok &= images(ctx, sc, 333, 187, 3, 32, bsdf)
    print("ALL IDENTICAL" if ok else "SOMETHING DIFFERS", flush=True)
    if os.environ.get("TIME", "1") == "1":
        for name in ("scene.xml", "bunny20.xml"):
            sc = load(name)
            ctx.upload_scene(*sc.buffers())
            W, H, spp = 1920, 1080, int(os.environ.get("SPP", "256"))
            ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
            for pipe in (capi.PIPE_WAVELOCAL, capi.PIPE_ORDERED):
                for rep in range(3):
                    ctx.clear_sum(); ctx.reset_stats()
                    ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=spp, pipeline=pipe)
                    st = ctx.stats()
                print("%s pipe %d: %.2f ms, %.1f Mrays/s (retraced %d, parked %d of %d rays)" % (name, pipe, st["total_ms"], st["rays"] / st["total_ms"] / 1e3, st["exact_retraces"], st["tree_parked"], st["rays"]), flush=True)
    ctx.close()
    sys.exit(0 if ok else 1)


main()
