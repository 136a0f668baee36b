"""First-light check on a GPU box: KATs, closest-hit parity, image parity, a timing. Scratch tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from metalpathtracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = ob.OracleScene(); sc.load_xml(os.path.join(ROOT, "assets/scene.xml")); sc.build_bvh(); buf = sc.buffers()
ctx = capi.Context(0)
ctx.upload_scene(*buf)
# KATs
seeds = np.array([0, 1, 2, 12345, 0xFFFFFFFF], np.uint32)
print("pcg", ctx.kat_pcg(seeds))
u = np.linspace(0, 1, 1000001, dtype=np.float32)[:-1]
s, c = ctx.kat_sincos(u)
so = np.empty_like(u); co = np.empty_like(u)
import ctypes as C
for i in range(0, u.size, 9973):
    a = C.c_float(); b = C.c_float(); ob.lib().orc_sincos_2pi(float(u[i]), C.byref(a), C.byref(b))
    assert a.value == s[i] and b.value == c[i], (i, a.value, s[i], b.value, c[i])
print("sincos ok; max err vs numpy", np.abs(s - np.sin(2*np.pi*u.astype(np.float64))).max())
# closest hit parity
rng = np.random.default_rng(1)
n = 20000
o = np.tile(np.array([0, 20, 50], np.float32), (n, 1)) + rng.normal(0, 1, (n, 3)).astype(np.float32)
d = rng.normal(0, 1, (n, 3)).astype(np.float32); d[:, 2] = -np.abs(d[:, 2]) - 0.5; d /= np.linalg.norm(d, axis=1, keepdims=True)
t, prim, nrm, front = ctx.trace_rays(o, d)
bad = 0
for i in range(0, n, 7):
    to, po, no, fo = ob.first_hit(o[i], d[i], buf)
    if not (po == prim[i] and (to == t[i] or (np.isinf(to) and np.isinf(t[i])))): bad += 1
print("closest-hit mismatches:", bad, "hits:", (prim >= 0).sum())
# image parity philox
W, H, spp = 256, 144, 8
uo = ob.make_uniforms(W, H, sc.prim_count, sc.triangle_count)
ug = capi.Uniforms.from_buffer_copy(bytes(uo))
ctx.resize(W, H); ctx.set_uniforms(ug)
for pipe in (capi.PIPE_WAVEFRONT, capi.PIPE_MEGAKERNEL):
    ctx.clear_sum(); ctx.reset_stats()
    ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=spp, pipeline=pipe, flags=capi.FLAG_COUNT_WORK)
    g = ctx.read_sum()
    ref, ct = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=spp, threads=8)
    diff = np.abs(g - ref)
    print("pipe", pipe, "philox maxdiff", diff.max(), "bit-exact", np.array_equal(g, ref), ctx.stats(), {k: ct[k] for k in ("rays","node_pops","aabb_pass","prim_tests","paths")})
# literal frame
ctx.resize(W, H)
uo = ob.make_uniforms(W, H, sc.prim_count, sc.triangle_count, random_seed=ob.host_seed_sequence(3), frame_count=1)
ctx.set_uniforms(capi.Uniforms.from_buffer_copy(bytes(uo)))
ctx.draw(rng_mode=capi.RNG_LITERAL, max_depth=32)
g = ctx.read_frame()
ref, ct = ob.render(uo, buf, rng_mode=ob.RNG_LITERAL, max_depth=32, accumulate=0, threads=8)
d2 = (g[..., :3] - ref[..., :3]).astype(np.float64)
print("literal frame: rms", np.sqrt((d2**2).sum(-1).mean()), "max", np.abs(d2).max(), "n>1e-3", (np.abs(d2).max(-1) > 1e-3).sum())
# timing 1080p
W, H = 1920, 1080
uo = ob.make_uniforms(W, H, sc.prim_count, sc.triangle_count)
ctx.resize(W, H); ctx.set_uniforms(capi.Uniforms.from_buffer_copy(bytes(uo)))
for pipe in (capi.PIPE_WAVEFRONT, capi.PIPE_MEGAKERNEL):
    for rep in range(2):
        ctx.clear_sum(); ctx.reset_stats()
        t0 = time.time()
        ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=64, pipeline=pipe)
        dt = time.time() - t0
        st = ctx.stats()
        print("pipe", pipe, "1080p x64: wall %.3fs total_ms %.1f trace_ms %.1f launches %d rays %d -> %.1f Mrays/s" % (dt, st["total_ms"], st["trace_kernel_ms"], st["trace_launches"], st["rays"], st["rays"]/st["total_ms"]/1e3))
