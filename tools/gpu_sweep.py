"""Interleaved A/B timing of env-configured variants in ONE process is not possible (env read at create), so this
spawns nothing: it creates several contexts with different env settings and alternates between them."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = os.environ.get("SCENE", "scene.xml")
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", scene), sc); assert st == 0
sc.buildBVH()
bufs = sc.buffers()
W, H = 1920, 1080
u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount())
variants = []
for spec in os.environ.get("VARIANTS", "MPT_LIGHT_BUDGET=8;MPT_LIGHT_BUDGET=12").split(";"):
    env = dict(kv.split("=") for kv in spec.split("|") if kv)
    for k, v in env.items(): os.environ[k] = v
    ctx = capi.Context(0); ctx.upload_scene(*bufs); ctx.resize(W, H); ctx.set_uniforms(u)
    for k in env: os.environ.pop(k)
    variants.append((spec, ctx))
spp = int(os.environ.get("SPP", "64")); depth = int(os.environ.get("DEPTH", "8")); pipe = int(os.environ.get("PIPE", "2"))
times = {s: [] for s, _ in variants}
for rnd in range(int(os.environ.get("ROUNDS", "5"))):
    for spec, ctx in variants:
        ctx.clear_sum(); ctx.reset_stats()
        ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=depth, sample_count=spp, pipeline=pipe)
        times[spec].append(ctx.stats()["total_ms"])
for spec, _ in variants:
    t = np.array(times[spec][1:])
    print("%-40s median %.2f ms  min %.2f ms" % (spec, np.median(t), t.min()))
