"""Minimal driver for rocprofv3 runs: scene.xml 1080p, SPP samples, one pipeline (scratch tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = os.environ.get("SCENE", "scene.xml")     # a file under assets/, or "config4" (the 1,000,003-primitive scene of tools/config4_scene.py)
if scene == "config4":
    import tempfile, config4_scene
    xml = config4_scene.write(tempfile.mkdtemp(prefix="mpt_cfg4_"))
else:
    xml = os.path.join(ROOT, "assets", scene)
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(xml, sc); assert st == 0
ctx = capi.Context(0); host.make_ready(ctx, sc, int(os.environ.get("BVH", "0")))   # 0 reference, 1 binned, 2 GPU through the host, 3 device build
W, H = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))
cam = dict(pos=(0.0, 1.0, 3.4), fwd=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), vfov=40.0) if os.environ.get("CAM") == "cornell" else None   # (bench.py's CORNELL_CAM)
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), cam=cam))
spp = int(os.environ.get("SPP", "16")); pipe = int(os.environ.get("PIPE", "1")); depth = int(os.environ.get("DEPTH", "8"))
shards = int(os.environ.get("SHARDS", "1")); bsdf = int(os.environ.get("BSDF", "0"))   # SHARDS=8: rank 0's 1/8 tile shard; BSDF 1 = Scatter.h
launch = "async" if os.environ.get("ASYNC") else "sync"   # ASYNC=1: mpt_render_async + mpt_wait, one render at a time — the kernel variant bench.py's timed steps run (k_wavelocal_corun)
import json
print("WORKLOAD " + json.dumps(dict(scene=scene, width=W, height=H, spp=spp, depth=depth, pipeline=pipe, bvh=int(os.environ.get("BVH", "0")),
                                    prims=sc.getPrimitiveCount(), shards=shards, bsdf=bsdf, launch=launch, env=capi.knob_env(),
                                    **capi.build_id())), flush=True)
for rep in range(int(os.environ.get("REPS", "2"))):
    ctx.clear_sum(); ctx.reset_stats()
    kw = dict(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_count=spp, pipeline=pipe, slots_per_iter=int(os.environ.get("SLOTS", "0")),
              shard_rank=0, shard_count=shards)
    if launch == "async":
        ctx.render_async(**kw); ctx.wait()
    else:
        ctx.render(**kw)
    st = ctx.stats()
    print("pipe %d spp %d depth %d: total_ms %.2f trace_ms %.2f launches %d rays %d -> %.1f Mrays/s" % (pipe, spp, depth, st["total_ms"], st["trace_kernel_ms"], st["trace_launches"], st["rays"], st["rays"] / st["total_ms"] / 1e3), flush=True)
