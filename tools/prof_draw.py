"""rocprofv3 driver: 30 mpt_draw frames at 1280x720 (literal and philox)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
sc.buildBVH()
ctx = capi.Context(0); ctx.upload_scene(*sc.buffers())
W, H = 1280, 720
ctx.resize(W, H)
seeds = host.host_seed_sequence(3)
mode = capi.RNG_LITERAL if os.environ.get("MODE", "literal") == "literal" else capi.RNG_PHILOX
for f in range(30):
    ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), random_seed=seeds, frame_count=f))
    ctx.draw(rng_mode=mode, max_depth=32, sample_begin=f, sample_count=1)
ctx.synchronize()
