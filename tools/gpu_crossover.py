"""Where does the closest-first pipeline overtake the reference-order one?  Scenes of 1..20 bunnies (4,970..99,362
primitives), both pipelines, reference and binned-SAH trees, 1920x1080 x 64 spp (scratch tool behind MPT_AUTO_ORDERED_PRIMS)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")
tmp = tempfile.mkdtemp()
grid = [(x, z) for z in (0, -16, -32, -48) for x in (0, 16, -16, 32, -32)]
ctx = capi.Context(0)
W, H, spp = 1920, 1080, 64
for n in (1, 2, 3, 4, 6, 8, 12, 20) if not os.environ.get("QUICK") else (1, 2, 3, 4, 6):
    xml = os.path.join(tmp, "b%d.xml" % n)
    with open(xml, "w") as f:
        f.write('<Scene>\n<Sphere position="0,-10000,0" radius="10000" albedo="0.8,0.8,0.8" emission="0,0,0" materialType="0" emissionPower="0" />\n')
        f.write('<Sphere position="0,60,-20" radius="10" albedo="0.0,0.0,0.0" emission="1.0,0.9,0.7" materialType="0" emissionPower="5" />\n')
        for x, z in grid[:n]:
            f.write('<Mesh file="%s/bunny.obj" position="%d,0,%d" scale="10.0" albedo="0.9,0.5,0.3" emission="0,0,0" materialType="0" emissionPower="0" />\n' % (ASSETS, x, z))
        f.write("</Scene>\n")
    # (device build: each pipeline on the tree the builder makes for it — leaves of <= 6 for the reference-order kernel, <= 2 for
    #  the closest-first one, whatever the scene's size)
    for mode, tag in ((host.BVH_REFERENCE_SWEEP, "reference"), (host.BVH_BINNED_CENTROID, "binned"), (host.BVH_DEVICE, "device")):
        sc = host.Scene(); st, log = host.SceneLoader.LoadSceneFromXML(xml, sc, ASSETS); assert st == 0, log
        if mode != host.BVH_DEVICE:
            host.make_ready(ctx, sc, mode)
        ctx.resize(W, H)
        ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
        out = []
        for pipe in (capi.PIPE_WAVELOCAL, capi.PIPE_ORDERED):
            if mode == host.BVH_DEVICE:
                os.environ["MPT_LBVH_LEAF"] = "6" if pipe == capi.PIPE_WAVELOCAL else "2"
                host.make_ready(ctx, sc, mode)
            best = 1e9
            for rep in range(3):
                ctx.clear_sum(); ctx.reset_stats()
                ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=spp, pipeline=pipe)
                best = min(best, ctx.stats()["total_ms"])
            out.append(best)
        print("%2d bunnies %6d prims %-9s tree: reference-order %.2f ms, closest-first %.2f ms  -> %s" % (
            n, sc.getPrimitiveCount(), tag, out[0], out[1], "closest-first" if out[1] < out[0] else "reference-order"), flush=True)
