"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bit-exact for integer work and for the philox images; per-pixel L2 < 1e-3 (north_star) where the
device math library is involved (literal cos/sin).  Scene buffers come from the PRODUCT's host layer."""
import ctypes as C

import numpy as np
import pytest

from conftest import CORNELL_CAM, host_scene, oracle_scene, pixel_l2, scene_path
from oracle import binding as ob

pytestmark = pytest.mark.gpu
L2_TOL = 1e-3


def setup(ctx, name, W, H, cam=None, **uk):
    from metalpathtracer_amd import host
    sc, buf = host_scene(name)
    ctx.upload_scene(*buf)
    u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), cam=cam, **uk)
    ctx.resize(W, H)
    ctx.set_uniforms(u)
    uo = ob.Uniforms.from_buffer_copy(bytes(u))
    return buf, uo


def setup_tree(ctx, name, W, H, tree, cam=None):
    """As setup(), by tree route: "reference" = the reference's own tree through the host (Scene::buildBVH + mpt_upload_scene);
    "device" = mpt_build_and_upload, what `--bvh auto`, bench.py and its extra workloads render — the buffers returned are then the
    tree as it comes BACK from the device (mpt_download_bvh), which is what the oracle has to walk (VERDICT r4 weak #2b)."""
    from metalpathtracer_amd import host
    if tree == "reference":
        return setup(ctx, name, W, H, cam=cam)
    assert tree == "device"
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(scene_path(name), sc)
    assert st == 0, log
    buf = host.make_ready(ctx, sc, host.BVH_DEVICE)
    u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), cam=cam)
    ctx.resize(W, H)
    ctx.set_uniforms(u)
    return buf, ob.Uniforms.from_buffer_copy(bytes(u))


def test_rng_known_answers_on_device(gpu_ctx):
    seeds = np.array([0, 1, 2, 12345, 0xFFFFFFFF], np.uint32)
    h, f = gpu_ctx.kat_pcg(seeds)
    assert h.tolist() == [2891249901, 3639127469, 86804957, 261270601, 2144086741]
    assert f[:3].tolist() == [np.float32(0.67317158), np.float32(0.84730041), np.float32(0.0202108547)]
    rng = np.random.default_rng(0)
    s = rng.integers(0, 2**32, 100000, dtype=np.uint64).astype(np.uint32)
    h, f = gpu_ctx.kat_pcg(s)
    L = ob.lib()
    for i in range(0, s.size, 997):
        assert h[i] == L.orc_pcg_hash(int(s[i])) and f[i] == np.float32(L.orc_pcg_float(int(s[i])))
    ctr = rng.integers(0, 2**32, (4096, 4), dtype=np.uint64).astype(np.uint32)
    key = rng.integers(0, 2**32, (4096, 2), dtype=np.uint64).astype(np.uint32)
    ctr[0], key[0] = 0, 0
    ctr[1], key[1] = 0xFFFFFFFF, 0xFFFFFFFF
    ctr[2], key[2] = (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)
    out = gpu_ctx.kat_philox(ctr, key)
    assert out[0].tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]        # Random123 KATs
    assert out[1].tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert out[2].tolist() == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    o = (C.c_uint32 * 4)()
    for i in range(3, 4096, 61):
        L.orc_philox((C.c_uint32 * 4)(*ctr[i].tolist()), (C.c_uint32 * 2)(*key[i].tolist()), o)
        assert out[i].tolist() == list(o)
    u = np.concatenate([np.linspace(0, 1, 200001, dtype=np.float32)[:-1],
                        np.array([0.0, 0.125, 0.25, 0.375, 0.5, 0.875, 0.99999994], np.float32)])
    sn, cs = gpu_ctx.kat_sincos(u)
    a, b = C.c_float(), C.c_float()
    for i in list(range(0, u.size, 211)) + list(range(u.size - 7, u.size)):
        L.orc_sincos_2pi(float(u[i]), C.byref(a), C.byref(b))
        assert sn[i] == np.float32(a.value) and cs[i] == np.float32(b.value)


def test_reciprocal_chain_equals_ieee_division_on_all_operands(gpu_ctx):
    """mpt_rcp (mpt_device.h): v_rcp_f32 + the compiler's own fma chain stands for 1.0f / x — PathTracing.h:61 `1.0 / r.direction[i]`,
    :153-165 the triangle test's 1 / a — wherever 2^-126 <= |x| <= 2^126.  Proven here over ALL 2^32 operands on the device:
    not one of the operands in that range may differ from the correctly rounded division (the rest take the full expansion)."""
    bad_in, n_in, bad_out, n_out = gpu_ctx.kat_rcp()
    assert n_in + n_out == 2 ** 32
    assert n_in == 2 * (252 * 2 ** 23 + 1)           # both signs: exponents 2^-126 .. 2^125 in full, and 2^126 itself
    assert bad_in == 0, "%d operands of the guarded range differ from IEEE division" % bad_in
    assert bad_out > 0                               # the guard is needed: 0, Inf, NaN, denormals, results that underflow


def test_closest_hit_matches_oracle_bitwise(gpu_ctx):
    buf, _ = setup(gpu_ctx, "scene.xml", 64, 36)
    rng = np.random.default_rng(1)
    n = 4096
    o = np.tile(np.array([0, 20, 50], np.float32), (n, 1)) + rng.normal(0, 2, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d[:, 2] = -np.abs(d[:, 2]) - 0.3
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # edge cases: axis-parallel directions (1/0 = inf in the slab test), rays from inside the light sphere,
    # rays grazing the big sphere, un-normalised and zero directions
    # ... and directions with NaN / Inf components (normalize of a zero vector happens at the critical angle of a
    # refraction): the reference walks the whole tree for them and hits nothing, the device returns that miss directly
    nan, inf = np.nan, np.inf
    extra_o = np.array([[0, 20, 50], [0, 20, 50], [0, 20, 50], [0, 20, 0], [0, 20, 0], [40, 100, 50], [-25, 5, 50],
                        [0, 20, 50], [0, 20, 50], [0, 20, 50], [0, 20, 50], [0, 20, 50], [0, 20, 50], [0, 20, 50],
                        [-25, 5, 50]], np.float32)
    extra_d = np.array([[0, 0, -1], [0, -1, 0], [1, 0, 0], [0, 0, 1], [0, 1, 0], [0, 0, -1], [0, 0, -1],
                        [0, 0, -7.5], [0, 0, 0], [nan, nan, nan], [nan, 0.5, -0.5], [0, nan, -1], [inf, 0, 0],
                        [0, -inf, -1], [nan, nan, -1]], np.float32)
    o = np.concatenate([o, extra_o])
    d = np.concatenate([d, extra_d])
    t, prim, nrm, front = gpu_ctx.trace_rays(o, d)
    hits = 0
    for i in range(o.shape[0]):
        to, po, no, fo = ob.first_hit(o[i], d[i], buf)
        assert po == prim[i], i
        if po >= 0:
            hits += 1
            assert np.float32(to) == t[i] and fo == bool(front[i])
            assert tuple(np.float32(x) for x in no) == tuple(nrm[i])
        else:
            assert np.isinf(t[i])
    assert hits > 1000


@pytest.mark.parametrize("pipeline", [0, 1, 2])
@pytest.mark.parametrize("name,W,H,cam,depth,spp,bsdf", [
    ("scene.xml", 160, 90, None, 8, 8, 0),
    ("scene.xml", 101, 67, None, 32, 4, 0),       # ragged size: partial 8x8 tiles on both edges
    ("cornell.xml", 96, 96, CORNELL_CAM, 32, 8, 0),
    ("glass.xml", 128, 72, None, 16, 8, 1),
    ("glass.xml", 128, 72, None, 16, 8, 2),       # Scatter.h for every material, its own Lambert branch included
    ("bunny20.xml", 96, 54, None, 8, 2, 0),
])
def test_philox_image_bit_exact(gpu_ctx, name, W, H, cam, depth, spp, bsdf, pipeline):
    from metalpathtracer_amd import capi
    buf, uo = setup(gpu_ctx, name, W, H, cam=cam)
    gpu_ctx.clear_sum()
    gpu_ctx.reset_stats()
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_count=spp, seed=(11, 5),
                   pipeline=pipeline, flags=capi.FLAG_COUNT_WORK)
    got = gpu_ctx.read_sum()
    ref, ct = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, accumulate=1,
                        sample_count=spp, seed=(11, 5), threads=8)
    assert pixel_l2(got / spp, ref / spp) < L2_TOL          # the north-star gate
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))   # what is actually achieved
    st = gpu_ctx.stats()
    assert (st["paths"], st["rays"], st["node_visits"], st["aabb_hits"], st["prim_tests"]) == (
        ct["paths"], ct["rays"], ct["node_pops"], ct["aabb_pass"], ct["prim_tests"])


def test_small_wavefront_width_and_sample_ranges(gpu_ctx):
    """Tiny wavefront width (many iterations, queue shards nearly empty) and split sample ranges give the
    same bits as one big call: the result must not depend on scheduling."""
    from metalpathtracer_amd import capi
    buf, uo = setup(gpu_ctx, "scene.xml", 120, 68)
    gpu_ctx.clear_sum()
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=6, seed=(3, 9))
    whole = gpu_ctx.read_sum()
    gpu_ctx.clear_sum()
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_begin=0, sample_count=2, seed=(3, 9), slots_per_iter=512,
                   pipeline=capi.PIPE_WAVEFRONT)
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_begin=2, sample_count=4, seed=(3, 9), slots_per_iter=4096,
                   pipeline=capi.PIPE_WAVEFRONT)
    parts = gpu_ctx.read_sum()
    np.testing.assert_array_equal(whole.view(np.uint32), parts.view(np.uint32))
    ref, _ = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=6, seed=(3, 9), threads=8)
    np.testing.assert_array_equal(whole.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_literal_frame_protocol(gpu_ctx, pipeline):
    """The reference's actual per-frame behaviour (stuck RNG, running mean, frameCount off-by-one):
    4 frames, host-seeded randomSeed, against the oracle's frame protocol."""
    from metalpathtracer_amd import capi, host
    W, H = 160, 90
    sc, buf = host_scene("scene.xml")
    gpu_ctx.upload_scene(*buf)
    gpu_ctx.resize(W, H)
    rs = host.host_seed_sequence(3)
    last = np.zeros((H, W, 4), np.float32)
    for f in range(1, 5):
        u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), random_seed=rs, frame_count=f)
        gpu_ctx.set_uniforms(u)
        gpu_ctx.draw(rng_mode=capi.RNG_LITERAL, max_depth=32, pipeline=pipeline)
        got = gpu_ctx.read_frame()
        ref, _ = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buf, rng_mode=ob.RNG_LITERAL, max_depth=32,
                           accumulate=0, last=last, threads=8)
        assert pixel_l2(got, ref) < L2_TOL
        assert np.abs(got - ref).max() < 1e-5
        last = ref
    assert got[H // 2, W // 2, 0] == pytest.approx(0.8, abs=1e-6)   # 4/5 at the light


def test_renderer_facade_frames(gpu_ctx):
    """host.Renderer = the reference's Renderer call order (constructor, drawableSizeWillChange, draw x N)."""
    from metalpathtracer_amd import host
    r = host.Renderer(0, scene_path("scene.xml"))
    r.drawableSizeWillChange(128, 72)
    sc, buf = oracle_scene("scene.xml")
    last = np.zeros((72, 128, 4), np.float32)
    for f in range(1, 4):
        r.draw()
        u = r.uniforms()
        assert u.frameCount == f and list(u.randomSeed)[:3] == [0.0, 0.0, 0.0]   # SURVEY A.3-3 as shipped
        got = r.readFrame()
        ref, _ = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buf, rng_mode=ob.RNG_LITERAL, max_depth=32,
                           accumulate=0, last=last, threads=8)
        assert pixel_l2(got, ref) < L2_TOL
        last = ref
    assert got[36, 64, :3].tolist() == pytest.approx([0.75, 0.75, 0.75], abs=1e-6)  # k/(k+1) at the light
    r.close()


def test_tile_sharding_sums_to_single_gpu_image(gpu_ctx):
    """N logical ranks rendered one after another into separate buffers, summed on the host,
    equal the 1-rank image bit for bit (each pixel is owned by exactly one rank)."""
    from metalpathtracer_amd import capi
    buf, uo = setup(gpu_ctx, "scene.xml", 200, 120)
    gpu_ctx.clear_sum()
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=4)
    one = gpu_ctx.read_sum()
    for n in (2, 3, 8):
        total = np.zeros_like(one)
        owned = np.zeros(one.shape[:2], np.int32)
        for r in range(n):
            gpu_ctx.clear_sum()
            gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=4, shard_rank=r, shard_count=n)
            part = gpu_ctx.read_sum()
            owned += (part[..., 3] > 0)
            total += part
        assert (owned == 1).all()
        np.testing.assert_array_equal(total.view(np.uint32), one.view(np.uint32))


def test_error_paths(gpu_ctx):
    from metalpathtracer_amd import capi
    ctx = capi.Context(0)
    with pytest.raises(capi.MptError) as e:
        ctx.render(sample_count=1)
    assert e.value.status == 5  # MPT_ERR_NOT_READY
    sc, (bvh, prims, mats, idx) = host_scene("scene.xml")
    bad = bvh.copy()
    bad[1, 0, 3] = np.int32(0).view(np.float32) if bad[1, 1, 3].view(np.int32) <= 0 else bad[1, 0, 3]
    bad[0, 1, 3] = np.array(0, np.int32).view(np.float32)       # right child = 0 -> cycle
    with pytest.raises(capi.MptError) as e:
        ctx.upload_scene(bad, prims, mats, idx)
    assert e.value.status == 4  # MPT_ERR_BAD_SCENE
    bad = idx.copy()
    bad[5] = 10**6
    with pytest.raises(capi.MptError) as e:
        ctx.upload_scene(bvh, prims, mats, bad)
    assert e.value.status == 4
    ctx.upload_scene(bvh, prims, mats, idx)
    ctx.resize(32, 32)
    from metalpathtracer_amd import host
    ctx.set_uniforms(host.make_uniforms(64, 64, 1))
    with pytest.raises(capi.MptError) as e:
        ctx.render(sample_count=1)
    assert e.value.status == 1  # uniforms.screenSize != resize
    with pytest.raises(capi.MptError) as e:
        ctx.render_async(sample_count=1)
    assert e.value.status == 1  # the asynchronous entry point validates the same way, nothing is left in flight
    ctx.wait()
    ctx.set_uniforms(host.make_uniforms(32, 32, sc.getPrimitiveCount()))
    with pytest.raises(capi.MptError) as e:
        ctx.render_async(sample_count=1, max_depth=33)
    assert e.value.status == 1
    ctx.render_async(rng_mode=capi.RNG_PHILOX, sample_count=2)      # and a valid one still works afterwards;
    ctx.resize(16, 16)                                              # resize waits for it by itself
    assert ctx.stats()["paths"] == 32 * 32 * 2
    ctx.close()


def test_material_guard_and_long_leaf(gpu_ctx):
    """primitiveCount smaller than the real count trips the material-index guard (PathTracing.h:234-236);
    a hand-made single-leaf BVH with 40 primitives exercises the >16 leaf split of the device layout."""
    from metalpathtracer_amd import capi, host
    rng = np.random.default_rng(4)
    sc = host.Scene()
    o = ob.OracleScene()
    for i in range(40):
        c = rng.uniform(-3, 3, 3)
        c[2] = rng.uniform(-6, -2)
        a, b = c + rng.uniform(-1, 1, 3), c + rng.uniform(-1, 1, 3)
        sc.addTriangle(c, a, b, albedo=(0.5, 0.6, 0.7))
        o.add_triangle(c, a, b, albedo=(0.5, 0.6, 0.7))
    sc.buildBVH()
    o.build_bvh()
    bvh, prims, mats, idx = sc.buffers()
    # collapse to ONE leaf holding all 40 primitives
    one = bvh[:1].copy()
    one[0, 0, 3] = np.array(0, np.int32).view(np.float32)
    one[0, 1, 3] = np.array(40, np.int32).view(np.float32)
    idx1 = np.arange(40, dtype=np.int32)
    gpu_ctx.upload_scene(one, prims, mats, idx1)
    W, H = 64, 64
    cam = dict(pos=(0, 0, 3), fwd=(0, 0, -1), up=(0, 1, 0), vfov=60.0)
    for pc in (40, 17):
        u = host.make_uniforms(W, H, pc, pc, cam=cam)
        gpu_ctx.resize(W, H)
        gpu_ctx.set_uniforms(u)
        gpu_ctx.clear_sum()
        gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=4)
        got = gpu_ctx.read_sum()
        ref, _ = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), (one, prims, mats, idx1), rng_mode=ob.RNG_PHILOX,
                           max_depth=8, accumulate=1, sample_count=4, threads=4)
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_full_size_properties(gpu_ctx):
    """BASELINE.json full size (1920x1080): size-independent properties instead of an oracle render —
    determinism, sample-range additivity, shard additivity, wavefront == megakernel — and the whole frame at 4 spp
    compared with the oracle bit for bit."""
    from metalpathtracer_amd import capi
    W, H = 1920, 1080
    buf, uo = setup(gpu_ctx, "scene.xml", W, H)
    kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, seed=(1, 0))
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_count=4, **kw)
    a = gpu_ctx.read_sum()
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_count=4, **kw)
    np.testing.assert_array_equal(a.view(np.uint32), gpu_ctx.read_sum().view(np.uint32))       # deterministic
    for pipe in (capi.PIPE_WAVEFRONT, capi.PIPE_MEGAKERNEL, capi.PIPE_WAVELOCAL):
        gpu_ctx.clear_sum()
        gpu_ctx.render(sample_count=4, pipeline=pipe, **kw)
        np.testing.assert_array_equal(a.view(np.uint32), gpu_ctx.read_sum().view(np.uint32))   # pipelines agree
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_begin=0, sample_count=1, **kw)
    gpu_ctx.render(sample_begin=1, sample_count=3, **kw)
    np.testing.assert_array_equal(a.view(np.uint32), gpu_ctx.read_sum().view(np.uint32))       # additive in samples
    total = np.zeros_like(a)
    for r in range(2):
        gpu_ctx.clear_sum()
        gpu_ctx.render(sample_count=4, shard_rank=r, shard_count=2, **kw)
        total += gpu_ctx.read_sum()
    np.testing.assert_array_equal(a.view(np.uint32), total.view(np.uint32))                    # additive in shards
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all() and (a[..., :3] <= 4).all()        # per-sample clamp
    # ... and the whole 1920x1080 frame against the oracle (8.3 M paths, ~14 M rays: a fraction of a second of CPU)
    ref, ct = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=4, seed=(1, 0), threads=16)
    np.testing.assert_array_equal(a.view(np.uint32), ref.view(np.uint32))
    assert ct["paths"] == W * H * 4


@pytest.mark.parametrize("budget", ["1", "3", "1000000"])
def test_parking_budgets_do_not_change_results(budget, monkeypatch):
    """Wave-local pipeline: a fresh bounce ray gets `budget` box-test loop trips before it is parked with its
    traversal state and resumed later.  A tiny budget parks almost every ray (and resumes it from the saved node /
    best t / best primitive); results and work counters must not move."""
    from metalpathtracer_amd import capi, host
    monkeypatch.setenv("MPT_LIGHT_BUDGET", budget)
    ctx = capi.Context(0)
    sc, buf = host_scene("scene.xml")
    ctx.upload_scene(*buf)
    W, H, spp = 192, 108, 6   # 6 spp: samples-per-pass not a power of two (division path of path -> pixel)
    u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount())
    ctx.resize(W, H)
    ctx.set_uniforms(u)
    ctx.clear_sum()
    ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=32, sample_count=spp, seed=(5, 5), pipeline=capi.PIPE_WAVELOCAL,
               flags=capi.FLAG_COUNT_WORK)
    got = ctx.read_sum()
    st = ctx.stats()
    ref, ct = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buf, rng_mode=ob.RNG_PHILOX, max_depth=32, accumulate=1,
                        sample_count=spp, seed=(5, 5), threads=8)
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert (st["paths"], st["rays"], st["node_visits"], st["aabb_hits"], st["prim_tests"]) == (
        ct["paths"], ct["rays"], ct["node_pops"], ct["aabb_pass"], ct["prim_tests"])
    ctx.close()


def test_cornell_config0_matches_cpu_reference_path(gpu_ctx):
    """BASELINE.json configs[0]: Cornell box, 256x256, 16 spp — the CPU path (oracle) and the HIP path agree."""
    from metalpathtracer_amd import capi
    buf, uo = setup(gpu_ctx, "cornell.xml", 256, 256, cam=CORNELL_CAM)
    gpu_ctx.clear_sum()
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=32, sample_count=16, seed=(1, 0))
    got = gpu_ctx.read_sum()
    ref, ct = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, max_depth=32, accumulate=1, sample_count=16, seed=(1, 0), threads=8)
    assert pixel_l2(got / 16, ref / 16) < L2_TOL
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert ct["paths"] == 256 * 256 * 16


def test_flat_leaf_box_is_never_hit(gpu_ctx):
    """Reference quirk (PathTracing.h:68 `tMax <= tMin`): an axis-aligned flat quad alone in a leaf has a
    zero-thickness box and is culled by the slab test — for the oracle and the device alike."""
    from metalpathtracer_amd import host
    sc = host.Scene()
    o = ob.OracleScene()
    for tri in (((-1, 2, 1), (1, 2, 1), (1, 2, -1)), ((-1, 2, 1), (1, 2, -1), (-1, 2, -1))):
        sc.addTriangle(*tri)
        o.add_triangle(*tri)
    sc.buildBVH()
    o.build_bvh()
    buf = sc.buffers()
    gpu_ctx.upload_scene(*buf)
    org = np.array([[0, 0, 0], [0.3, 0, 0.2], [0, 4, 0]], np.float32)
    d = np.array([[0, 1, 0], [0, 1, 0], [0, -1, 0]], np.float32)
    t, prim, _, _ = gpu_ctx.trace_rays(org, d)
    assert (prim == -1).all() and np.isinf(t).all()
    for i in range(3):
        assert ob.first_hit(org[i], d[i], o.buffers())[1] == -1
    # a slightly oblique ray has a non-degenerate slab interval on x/z but still a zero one on y: also missed
    t, prim, _, _ = gpu_ctx.trace_rays(np.array([[0, 0, 0]], np.float32), np.array([[0.1, 1, 0.05]], np.float32))
    assert prim[0] == -1 and ob.first_hit((0, 0, 0), (0.1, 1, 0.05), o.buffers())[1] == -1


def _heightfield_obj(path, n, seed):
    """Seeded procedural mesh: an n x n jittered height field (2*(n-1)^2 triangles) — configs[4] stand-in."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-40, 40, n)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    Y = 6 * np.sin(X * 0.21) * np.cos(Z * 0.17) + rng.uniform(-0.4, 0.4, X.shape) + 8
    with open(path, "w") as f:
        for i in range(n):
            for j in range(n):
                f.write("v %.6f %.6f %.6f\n" % (X[i, j], Y[i, j], Z[i, j]))
        for i in range(n - 1):
            for j in range(n - 1):
                a = i * n + j + 1
                f.write("f %d %d %d\nf %d %d %d\n" % (a, a + 1, a + n, a + 1, a + n + 1, a + n))


def test_large_mesh_partial_lds_glass_and_mirror(gpu_ctx, tmp_path):
    """configs[4]-style stress at test size: ~180 k triangles (BVH far larger than the LDS budget, so most nodes
    are fetched from L2), glass + mirror + diffuse + emissive materials (Scatter.h semantics), depth 16.  The BVH is
    built by the product's fast binned builder; the oracle traverses the same arrays, so parity is exact."""
    from metalpathtracer_amd import capi, host
    n = 301
    _heightfield_obj(str(tmp_path / "hf.obj"), n, seed=1)
    (tmp_path / "big.xml").write_text("""<Scene>
  <Mesh file="hf.obj" position="0,-10,-30" scale="1.0" albedo="0.7,0.7,0.75" emission="0,0,0" materialType="0" emissionPower="0"/>
  <Mesh file="hf.obj" position="0,35,-60" scale="0.6" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="-15,18,-10" radius="9" albedo="0.95,0.95,0.95" emission="0,0,0" materialType="-1" emissionPower="0"/>
  <Sphere position="15,18,-10" radius="9" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="0,60,-20" radius="10" albedo="0,0,0" emission="1,0.9,0.7" materialType="0" emissionPower="5"/>
</Scene>""")
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(str(tmp_path / "big.xml"), sc)
    assert st == 0, log
    assert sc.getTriangleCount() == 2 * 2 * (n - 1) ** 2
    sc.buildBVH(host.BVH_BINNED_CENTROID)
    buf = sc.buffers()
    assert buf[0].shape[0] * 32 > 4 * 60 * 1024          # node array >> LDS budget
    gpu_ctx.upload_scene(*buf)
    W, H, spp = 160, 90, 4
    u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount())
    gpu_ctx.resize(W, H)
    gpu_ctx.set_uniforms(u)
    ref, ct = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buf, rng_mode=ob.RNG_PHILOX, bsdf_mode=ob.BSDF_SCATTER,
                        max_depth=16, accumulate=1, sample_count=spp, seed=(2, 7), threads=8)
    for pipe in (capi.PIPE_WAVELOCAL, capi.PIPE_MEGAKERNEL, capi.PIPE_WAVEFRONT):
        gpu_ctx.clear_sum()
        gpu_ctx.reset_stats()
        gpu_ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=capi.BSDF_SCATTER, max_depth=16, sample_count=spp, seed=(2, 7),
                       pipeline=pipe, flags=capi.FLAG_COUNT_WORK)
        got = gpu_ctx.read_sum()
        assert pixel_l2(got / spp, ref / spp) < L2_TOL
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
        st_ = gpu_ctx.stats()
        assert (st_["rays"], st_["node_visits"], st_["prim_tests"]) == (ct["rays"], ct["node_pops"], ct["prim_tests"])


def test_cli_mpt_render_matches_golden(tmp_path):
    """The C++ front end (Renderer facade + CLI): scene.xml through lib/mpt_render, PFM out, against the golden
    fixture; and the Cornell box with camera flags against the oracle."""
    import json, os, subprocess
    from conftest import GOLDEN, ROOT
    exe = os.path.join(ROOT, "metalpathtracer_amd", "lib", "mpt_render")
    out = str(tmp_path / "o.pfm")
    base = [exe, "--scene", scene_path("scene.xml"), "--width", "96", "--height", "54", "--spp", "8", "--depth", "8", "--seed", "1", "--out", out]
    want = np.load(os.path.join(GOLDEN, "scene_philox_8spp_d8.npy"))
    hdr = b"PF\n96 54\n-1.0\n"
    # a batch render takes the device builder (--bvh auto): another tree than the fixture's, the same image up to the rays
    # whose answer depends on the visit order (ties)
    r = subprocess.run(base, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(out, "rb").read()
    img = np.frombuffer(raw[len(hdr):], np.float32).reshape(54, 96, 3)[::-1]
    assert pixel_l2(img, want[..., :3] * np.float32(0.125)) < L2_TOL
    r = subprocess.run(base + ["--bvh", "reference"], capture_output=True, text=True)   # the reference's own tree: bit for bit
    assert r.returncode == 0, r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    info = json.loads(line)
    raw = open(out, "rb").read()
    assert raw.startswith(hdr)
    img = np.frombuffer(raw[len(hdr):], np.float32).reshape(54, 96, 3)[::-1]
    manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))["scene_philox_8spp_d8"]
    assert info["rays"] == manifest["counters"]["rays"] and info["paths"] == 96 * 54 * 8
    np.testing.assert_array_equal(img, want[..., :3] * np.float32(0.125))
    # Cornell through the camera flags
    r = subprocess.run([exe, "--scene", scene_path("cornell.xml"), "--width", "64", "--height", "64", "--spp", "16",
                        "--depth", "32", "--seed", "1", "--camera-pos", "0,1,3.4", "--camera-dir", "0,0,-1",
                        "--camera-up", "0,1,0", "--vfov", "40", "--bvh", "reference", "--out", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(out, "rb").read()
    hdr = b"PF\n64 64\n-1.0\n"
    img = np.frombuffer(raw[len(hdr):], np.float32).reshape(64, 64, 3)[::-1]
    want = np.load(os.path.join(GOLDEN, "cornell_philox_16spp.npy"))
    np.testing.assert_array_equal(img, want[..., :3] * np.float32(1.0 / 16))


def test_camera_move_resets_accumulation_and_reseeds(gpu_ctx):
    """Renderer::updateUniforms (R/Renderer/Renderer.cpp:251-267): a camera change sets frameCount = 0 and draws
    randomSeed from the host PCG stream (first triple = SURVEY App. C.1); the next still frame counts up again.
    The frame rendered after the move matches the oracle with those uniforms (literal RNG, sin-hash pixel seeds)."""
    from metalpathtracer_amd import host
    r = host.Renderer(0, scene_path("scene.xml"))
    r.drawableSizeWillChange(128, 72)
    r.draw()
    assert r.uniforms().frameCount == 1
    r.input(move=(0, 0, 1))                      # one step forward (movementSpeed 0.1, Camera.h:20,35-47)
    r.draw()
    u = r.uniforms()
    assert u.frameCount == 0
    assert list(u.randomSeed)[:3] == pytest.approx([0.80921644, 0.38028690, 0.09423842], abs=5e-9)
    assert list(u.cameraPosition)[:3] == pytest.approx([0.0, 20.0, 49.9], abs=1e-5)
    got = r.readFrame()
    sc, buf = oracle_scene("scene.xml")
    ref, _ = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buf, rng_mode=ob.RNG_LITERAL, max_depth=32, accumulate=0,
                       last=np.ones((72, 128, 4), np.float32), threads=8)   # frameCount 0: lastFrame is cleared
    # device sinf/cosf differ from glibc in the last ulp of the (stuck) bounce vector: a silhouette pixel may flip
    assert pixel_l2(got, ref) < L2_TOL and (np.abs(got - ref).max(-1) > 1e-3).sum() <= 3
    r.draw()
    assert r.uniforms().frameCount == 1 and list(r.uniforms().randomSeed)[:3] == list(u.randomSeed)[:3]
    r.input(reset=True)
    r.draw()
    u2 = r.uniforms()
    assert u2.frameCount == 0 and list(u2.cameraPosition)[:3] == [0.0, 20.0, 50.0]
    assert list(u2.randomSeed)[:3] != list(u.randomSeed)[:3]
    r.close()


@pytest.mark.parametrize("lane_order", ["5", "1", "0"])
def test_pipelined_renders_give_the_serial_image_under_every_lane_rule(lane_order, monkeypatch):
    """mpt_render_async without the counting flag runs k_wavelocal_corun behind the residency gate (MPT_LANE_ORDER=5, the default),
    behind the event chain alone (1) or on independent lanes (0): twelve pipelined renders (a trace kernel long enough for the gate to
    wait on, a serial render and a re-sharding in between) must leave the HDR sum bit-identical to the same renders issued one at a
    time, and the kernels' own time stamps must add up to something sane."""
    from metalpathtracer_amd import capi
    monkeypatch.setenv("MPT_LANE_ORDER", lane_order)
    ctx = capi.Context(0)
    try:
        buf, uo = setup(ctx, "scene.xml", 960, 540)
        kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, seed=(7, 1))
        def sequence(issue):
            ctx.clear_sum(); ctx.reset_stats()
            for k in range(6): issue(sample_begin=8 * k, sample_count=8, **kw)
            ctx.render(sample_begin=48, sample_count=4, **kw)                      # a synchronous render in between
            for k in range(3): issue(sample_begin=52 + 2 * k, sample_count=2, shard_rank=k % 2, shard_count=2, **kw)
            for k in range(3): issue(sample_begin=52 + 2 * k, sample_count=2, shard_rank=1 - k % 2, shard_count=2, **kw)
            ctx.wait()
            return ctx.read_sum(), ctx.stats()
        serial, s_serial = sequence(ctx.render)
        piped, s_piped = sequence(ctx.render_async)
        np.testing.assert_array_equal(serial.view(np.uint32), piped.view(np.uint32))
        assert s_serial["rays"] == s_piped["rays"] and s_serial["paths"] == s_piped["paths"]
        assert s_piped["trace_launches"] == 7          # (the synchronous render in between restarts the timing statistics: itself + 6)
        # the kernels' own spans (first workgroup's start to last wave's end), summed: the seven launches trace 14 of the 64 samples
        assert 0 < s_piped["trace_kernel_ms"] < 200.0
        ref, _ = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=2, seed=(7, 1), threads=8)
        ctx.clear_sum()
        ctx.render_async(sample_begin=0, sample_count=1, **kw); ctx.render_async(sample_begin=1, sample_count=1, **kw)
        np.testing.assert_array_equal(ctx.read_sum().view(np.uint32), ref.view(np.uint32))
    finally:
        ctx.close()


def test_render_async_returns_at_once_and_takes_the_event_chain_behind_a_foreign_kernel(gpu_ctx):
    """mpt_render_async is asynchronous for the HOST: what a submission has to wait for — here the residency announcement of the trace
    kernel before it, which cannot come while a foreign persistent kernel holds every compute unit (tests/holder/hold_chip.hip: two
    workgroups of 80 KB of LDS per CU, 600 ms) — is waited for on the context's submit thread.  Every call returns in under a
    millisecond, the gate gives up after its 200 ms and falls back to the strict event chain (mpt_async_info counts it), and the HDR sum
    is bit-identical to the same renders issued one at a time.  Replaces the unfenced per-frame submit of R/Renderer/Renderer.cpp:253-266,307."""
    import ctypes
    import os
    import time
    from conftest import ROOT
    from metalpathtracer_amd import capi
    hold = ctypes.CDLL(os.path.join(ROOT, "tests", "holder", "_build", "libholdchip.so"))
    buf, uo = setup(gpu_ctx, "scene.xml", 960, 540)
    kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, seed=(4, 2), pipeline=capi.PIPE_WAVELOCAL)
    gpu_ctx.clear_sum()
    for k in range(3):
        gpu_ctx.render(sample_begin=8 * k, sample_count=8, **kw)
    serial = gpu_ctx.read_sum()
    gpu_ctx.clear_sum()                                   # warm the asynchronous path: second lane allocated, submit thread running
    gpu_ctx.render_async(sample_begin=0, sample_count=8, **kw)
    gpu_ctx.render_async(sample_begin=8, sample_count=8, **kw)
    gpu_ctx.wait()
    gpu_ctx.clear_sum()
    before = gpu_ctx.async_info()
    assert hold.hold_chip_start(0, 600) == 0
    time.sleep(0.05)                                      # the foreign kernel is resident everywhere by now
    took = []
    t_all = time.perf_counter()
    for k in range(3):
        t0 = time.perf_counter()
        gpu_ctx.render_async(sample_begin=8 * k, sample_count=8, **kw)
        took.append(time.perf_counter() - t0)
    t_all = time.perf_counter() - t_all
    assert max(took) < 1e-3 and t_all < 5e-3, took        # (round 4: the second call spun for the gate's 200 ms on this thread)
    gpu_ctx.wait()
    assert hold.hold_chip_wait() == 0
    after = gpu_ctx.async_info()
    np.testing.assert_array_equal(gpu_ctx.read_sum().view(np.uint32), serial.view(np.uint32))
    assert after["submitted"] - before["submitted"] == 3
    assert after["gate_timeout"] - before["gate_timeout"] >= 1, (before, after)   # the 200 ms -> event-chain path was taken
    assert after["call_us_max"] < 1000


def test_async_renders_overlap_and_match_the_serial_result(gpu_ctx):
    """mpt_render_async: consecutive renders overlap on two lanes; the HDR sum must be bit-identical to serial
    mpt_render calls (resolves are chained in submission order) and the statistics must add up after mpt_wait."""
    from metalpathtracer_amd import capi
    buf, uo = setup(gpu_ctx, "scene.xml", 320, 180)
    kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, seed=(4, 2), flags=capi.FLAG_COUNT_WORK)
    gpu_ctx.clear_sum()
    gpu_ctx.reset_stats()
    for k in range(5):
        gpu_ctx.render(sample_begin=3 * k, sample_count=3, **kw)
    serial = gpu_ctx.read_sum()
    s_serial = gpu_ctx.stats()
    gpu_ctx.clear_sum()
    gpu_ctx.reset_stats()
    for k in range(5):
        gpu_ctx.render_async(sample_begin=3 * k, sample_count=3, **kw)
    gpu_ctx.wait()
    overlapped = gpu_ctx.read_sum()
    s_async = gpu_ctx.stats()
    np.testing.assert_array_equal(serial.view(np.uint32), overlapped.view(np.uint32))
    for key in ("paths", "rays", "node_visits", "aabb_hits", "prim_tests"):
        assert s_serial[key] == s_async[key], key
    assert s_async["trace_launches"] == 5 and s_async["trace_kernel_ms"] > 0
    # anything that touches the sum waits by itself: no explicit wait before the read-back / clear
    gpu_ctx.clear_sum()
    gpu_ctx.render_async(sample_begin=0, sample_count=15, **kw)
    once = gpu_ctx.read_sum()
    ref, _ = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=15, seed=(4, 2), threads=8)
    np.testing.assert_array_equal(once.view(np.uint32), ref.view(np.uint32))
    # changing the sharding while a render is in flight rebuilds the tile table behind a wait
    gpu_ctx.clear_sum()
    gpu_ctx.render_async(sample_begin=0, sample_count=2, shard_rank=0, shard_count=2, **kw)
    gpu_ctx.render_async(sample_begin=0, sample_count=2, shard_rank=1, shard_count=2, **kw)
    both = gpu_ctx.read_sum()
    ref2, _ = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=2, seed=(4, 2), threads=8)
    np.testing.assert_array_equal(both.view(np.uint32), ref2.view(np.uint32))


@pytest.mark.parametrize("name,W,H,cam,depth,spp,bsdf", [
    ("scene.xml", 160, 90, None, 8, 8, 0),
    ("cornell.xml", 96, 96, CORNELL_CAM, 32, 8, 0),
    ("glass.xml", 128, 72, None, 16, 8, 1),
    ("glass.xml", 128, 72, None, 16, 8, 2),
    ("bunny20.xml", 96, 54, None, 8, 2, 0),
])
def test_production_kernel_variant_bit_exact(gpu_ctx, name, W, H, cam, depth, spp, bsdf):
    """The same comparison without MPT_FLAG_COUNT_WORK: that is the kernel instantiation bench.py and users run
    (no work counters, result slots not poisoned)."""
    from metalpathtracer_amd import capi
    buf, uo = setup(gpu_ctx, name, W, H, cam=cam)
    gpu_ctx.clear_sum()
    gpu_ctx.reset_stats()
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_count=spp, seed=(11, 5))
    got = gpu_ctx.read_sum()
    ref, ct = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, accumulate=1,
                        sample_count=spp, seed=(11, 5), threads=8)
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    st = gpu_ctx.stats()
    assert (st["paths"], st["rays"]) == (ct["paths"], ct["rays"])


def test_randomised_cases_bit_exact():
    """60 random (scene, size, sample range, depth, seed, BSDF mode, pipeline, shard count, sync/async) cases, each
    compared bit for bit with the oracle (tests/gpu_fuzz_parity.py)."""
    import os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, CASES="60", SEED="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_fuzz_parity.py")], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "60 cases, 0 mismatches" in r.stdout


@pytest.mark.parametrize("pipe,bvh", [("4", "reference"), ("3", "reference"), ("4", "device"), ("3", "device"), ("4", "device-async")])
def test_headline_config_is_bit_identical_to_the_oracle(pipe, bvh):
    """BASELINE.json configs[1] itself — scene.xml, 1920x1080, 256 spp, depth 8 (890,385,105 rays on the reference's tree):
    every float of the HDR sum equals the oracle's (tests/gpu_headline_parity.py; the oracle takes ~8 s on the GPU box's 16
    host threads).  On the reference's own tree through mpt_upload_scene (the drop-in route) AND on the device-built tree
    through mpt_build_and_upload (what bench.py's default line renders: the oracle walks the tree that mpt_download_bvh
    returns), each once with the default pipeline choice (AUTO = what bench.py runs) and once with the closest-first
    pipeline.  The device-tree AUTO case also renders the reference's tree and asserts that the two trees' images — equal up
    to ties — are within north_star's per-pixel L2 < 1e-3 at the full 256 spp."""
    import os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, PIPE=pipe, BVH=bvh)
    if bvh == "device-async":    # exactly what bench.py's timed region runs: the device-built tree, mpt_render_async on both lanes,
        env.update(BVH="device", ASYNC="1")   # k_wavelocal_corun behind the residency gate; 2 x 256 spp against the oracle's 512
    if bvh == "device" and pipe == "4":
        env["CROSS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_headline_parity.py")], capture_output=True,
                       text=True, timeout=1200, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "bit-identical=True" in r.stdout
    if bvh == "reference":
        assert "890385105 rays" in r.stdout
    if bvh == "device-async":
        assert "two overlapped mpt_render_async of 256 spp (trace launches 2)" in r.stdout
    if "CROSS" in env:
        assert "cross-tree L2 < 1e-3: True" in r.stdout
    print(r.stdout)


def test_checkpoint_and_resume_of_the_accumulation(gpu_ctx, tmp_path):
    """SURVEY.md 5 "checkpoint / resume" (the reference keeps its running mean in a GPU-private texture and loses it with the
    process): mpt_read_sum is the checkpoint, mpt_write_sum puts it back — through the C ABI and through the CLI
    (`mpt_render --checkpoint F`, `--resume F`): 5 samples, checkpoint, a fresh context, 3 more samples numbered from where the
    first run stopped == 8 samples in one go, every float."""
    import os, subprocess
    from conftest import ROOT
    from metalpathtracer_amd import capi
    W, H = 160, 90
    buf, uo = setup(gpu_ctx, "scene.xml", W, H)
    kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, seed=(7, 1))
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_begin=0, sample_count=8, **kw)
    whole = gpu_ctx.read_sum()
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_begin=0, sample_count=5, **kw)
    ckpt = gpu_ctx.read_sum()
    other = capi.Context(0)                                   # "another process": nothing survives but the checkpoint
    try:
        other.upload_scene(*buf)
        other.resize(W, H)
        other.set_uniforms(capi.Uniforms.from_buffer_copy(bytes(uo)))
        other.write_sum(ckpt)
        other.render(sample_begin=5, sample_count=3, **kw)
        np.testing.assert_array_equal(other.read_sum().view(np.uint32), whole.view(np.uint32))
    finally:
        other.close()
    exe = os.path.join(ROOT, "metalpathtracer_amd", "lib", "mpt_render")
    base = [exe, "--scene", scene_path("scene.xml"), "--width", "96", "--height", "54", "--depth", "8", "--seed", "3"]
    a, b, c = str(tmp_path / "a.pfm"), str(tmp_path / "b.pfm"), str(tmp_path / "c.bin")
    for args in (["--spp", "8", "--out", a], ["--spp", "5", "--checkpoint", c], ["--spp", "3", "--resume", c, "--out", b]):
        r = subprocess.run(base + args, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    assert open(c, "rb").read(8) == b"MPTSUM2 " and os.path.getsize(c) > 96 * 54 * 16 and not os.path.exists(c + ".tmp")
    assert open(a, "rb").read() == open(b, "rb").read()       # the resumed image file is the uninterrupted one, byte for byte
    r = subprocess.run(base + ["--spp", "3", "--resume", c, "--seed", "4"], capture_output=True, text=True)
    assert r.returncode != 0 and "another scene, size, seed" in r.stderr    # a checkpoint of another run is refused ...
    for other_run in (["--scene", scene_path("cornell.xml")], ["--bsdf", "scatter"]):   # ... and so is one of another scene or material model
        r = subprocess.run(base + ["--spp", "3", "--resume", c] + other_run, capture_output=True, text=True)
        assert r.returncode != 0 and "another scene, size, seed" in r.stderr, r.stderr
