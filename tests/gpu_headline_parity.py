"""The headline configuration itself (scene.xml, 1920x1080, 256 spp, depth 8, philox seed (1,0)): HIP image vs the CPU
oracle, every float of the HDR sum compared bit for bit (the oracle needs ~8 s on 16 threads for its 890 M rays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
from oracle import binding as ob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
sc.buildBVH(); buf = sc.buffers()
W, H, spp = 1920, 1080, int(os.environ.get("SPP", "256"))
u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount())
ctx = capi.Context(0); ctx.upload_scene(*buf); ctx.resize(W, H); ctx.set_uniforms(u); ctx.clear_sum(); ctx.reset_stats()
SCENE_PIPE = int(os.environ.get("PIPE", str(capi.DEFAULT_PIPELINE)))
ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=spp, seed=(1, 0), pipeline=SCENE_PIPE)
got = ctx.read_sum(); s = ctx.stats()
t0 = time.time()
ref, ct = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1,
                    sample_count=spp, seed=(1, 0), threads=int(os.environ.get("THREADS", "16")))
dt = time.time() - t0
same = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
l2 = float(np.sqrt(np.mean(np.sum((got[..., :3] / spp - ref[..., :3] / spp) ** 2, -1))))
print("pipeline %d, re-traced in reference order %d, parked %d" % (SCENE_PIPE, s["exact_retraces"], s["tree_parked"]))
print("GPU %.1f ms (%d rays), oracle %.1f s (%d rays): bit-identical=%s, per-pixel L2 = %.3g, differing floats = %d" % (
    s["total_ms"], s["rays"], dt, ct["rays"], same, l2, int((got != ref).sum())))
sys.exit(0 if same and s["rays"] == ct["rays"] else 1)
