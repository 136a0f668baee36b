"""GPU parity tests of the closest-first pipeline (MPT_PIPE_ORDERED, the default) and of the code paths the big
BASELINE.json configs depend on: the multi-pass loop, the 1 M-triangle scene, bunny x20 at full size.

Everything is compared with the CPU oracle (the reference-order walk) bit for bit; where the oracle would take too long
(full-size renders) the reference-order HIP pipeline — itself oracle-checked in test_gpu_parity.py — stands in."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import CORNELL_CAM, host_scene
from oracle import binding as ob
from test_gpu_parity import _heightfield_obj, setup, setup_tree

pytestmark = pytest.mark.gpu


def _same(a, b):
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("count_work", [0, 1])
@pytest.mark.parametrize("name,W,H,cam,depth,spp,bsdf", [
    ("scene.xml", 160, 90, None, 8, 8, 0),
    ("scene.xml", 101, 67, None, 32, 4, 0),       # ragged size: partial 8x8 tiles on both edges
    ("cornell.xml", 96, 96, CORNELL_CAM, 32, 8, 0),
    ("glass.xml", 128, 72, None, 16, 8, 1),
    ("glass.xml", 128, 72, None, 16, 8, 2),
    ("bunny20.xml", 96, 54, None, 8, 2, 0),
])
def test_ordered_image_bit_exact(gpu_ctx, name, W, H, cam, depth, spp, bsdf, count_work):
    from metalpathtracer_amd import capi
    buf, uo = setup(gpu_ctx, name, W, H, cam=cam)
    assert gpu_ctx.accel_info()["ordered_ok"] == 1
    gpu_ctx.clear_sum()
    gpu_ctx.reset_stats()
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_count=spp, seed=(11, 5),
                   pipeline=capi.PIPE_ORDERED, flags=capi.FLAG_COUNT_WORK if count_work else 0)
    got = gpu_ctx.read_sum()
    ref, ct = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, accumulate=1,
                        sample_count=spp, seed=(11, 5), threads=8)
    _same(got, ref)
    st = gpu_ctx.stats()
    assert (st["paths"], st["rays"]) == (ct["paths"], ct["rays"])
    if count_work:      # the own tree must not visit more boxes than the reference's unordered walk (4 boxes per own node)
        assert st["prim_tests"] > 0 and st["node_visits"] * 4 < 2 * ct["node_pops"] + 8 * ct["rays"]


def test_ordered_closest_hit_matches_the_reference_walk(gpu_ctx):
    """262 k random rays per scene (through, around and inside the geometry; 1/64 with a zero direction component)
    through mpt_trace_rays_ordered vs mpt_trace_rays (reference-order walk, oracle-checked in test_gpu_parity.py) and a
    sample of them vs the oracle itself; the re-trace flags must be rare."""
    for name, lo, hi in (("scene.xml", 1e-5, 5e-2), ("bunny20.xml", 0, 1e-2), ("glass.xml", 1e-5, 5e-2)):
        sc, buf = host_scene(name)
        gpu_ctx.upload_scene(*buf)
        rng = np.random.default_rng(5)
        n = 1 << 18
        o = (rng.normal(size=(n, 3)) * [25, 10, 25] + [0, 12, 10]).astype(np.float32)
        tgt = rng.normal(size=(n, 3)) * [10, 8, 10] + [0, 6, 0]
        d = (tgt - o).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[: n // 64, rng.integers(0, 3)] = 0.0
        o[n // 2:] = o[n // 2:] * [0.2, 0.5, 0.2]      # origins inside the mesh / spheres as well
        t0, p0, n0, f0 = gpu_ctx.trace_rays(o, d)
        t1, p1, n1, f1, fl = gpu_ctx.trace_rays_ordered(o, d)
        _same(t0, t1)
        np.testing.assert_array_equal(p0, p1)
        _same(n0, n1)
        np.testing.assert_array_equal(f0, f1)
        assert (p0 >= 0).sum() > n // 10
        frac = float((fl[n // 64:] != 0).mean())
        assert lo <= frac <= hi, (name, frac)
        assert (fl[: n // 64] & 1).all()                # degenerate directions are flagged
        for i in rng.integers(0, n, 300):
            to, po, no, fo = ob.first_hit(o[i], d[i], buf)
            assert po == p1[i] and (po < 0 or (np.float32(to) == t1[i] and fo == bool(f1[i])))


def test_ordered_falls_back_when_the_scene_does_not_qualify(gpu_ctx):
    """More than 16 spheres (no always list) -> accel_info says so and MPT_PIPE_ORDERED renders through the
    reference-order pipeline: same image."""
    from metalpathtracer_amd import capi, host
    sc = host.Scene()
    rng = np.random.default_rng(2)
    for k in range(40):
        c = rng.uniform(-8, 8, 3).astype(np.float32) + np.array([0, 20, 20], np.float32)
        sc.addSphere([float(x) for x in c], 1.0 + 0.05 * k, albedo=(0.7, 0.6, 0.5))
    sc.buildBVH()
    buf = sc.buffers()
    gpu_ctx.upload_scene(*buf)
    assert gpu_ctx.accel_info()["ordered_ok"] == 0
    W, H = 96, 54
    u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount())
    gpu_ctx.resize(W, H)
    gpu_ctx.set_uniforms(u)
    gpu_ctx.clear_sum()
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=3, pipeline=capi.PIPE_ORDERED)
    got = gpu_ctx.read_sum()
    ref, _ = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1,
                       sample_count=3, seed=(1, 0), threads=4)
    _same(got, ref)
    with pytest.raises(capi.MptError):
        gpu_ctx.trace_rays_ordered(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32))


@pytest.mark.parametrize("pipeline", [2, 3])
def test_multi_pass_loop_matches_the_oracle(gpu_ctx, pipeline, monkeypatch):
    """MPT_PASS_SPP=3: a 10-spp render becomes passes of 3+3+3+1 samples (S not a power of two: the division path of
    path_to_pixel), slots / descriptor / cursors reused in-stream; serial and overlapped on both render lanes."""
    from metalpathtracer_amd import capi
    monkeypatch.setenv("MPT_PASS_SPP", "3")
    W, H, spp = 173, 99, 10
    buf, uo = setup(gpu_ctx, "scene.xml", W, H)
    kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, seed=(3, 9), pipeline=pipeline)
    ref, ct = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=spp, seed=(3, 9), threads=8)
    gpu_ctx.clear_sum()
    gpu_ctx.reset_stats()
    gpu_ctx.render(sample_count=spp, **kw)
    _same(gpu_ctx.read_sum(), ref)
    st = gpu_ctx.stats()
    assert st["trace_launches"] == 4 and st["rays"] == ct["rays"]
    gpu_ctx.clear_sum()                                  # two overlapped renders of 5 samples each: 3+2 passes per lane
    gpu_ctx.render_async(sample_begin=0, sample_count=5, **kw)
    gpu_ctx.render_async(sample_begin=5, sample_count=5, **kw)
    gpu_ctx.wait()
    _same(gpu_ctx.read_sum(), ref)


@pytest.fixture(scope="module")
def million_triangle_scene(tmp_path_factory):
    """BASELINE.json configs[4]: two 500 k-triangle jittered height fields (one of them glass), mirror + glass spheres,
    an emitter — 1,000,003 primitives (the generator tools/gpu_configs.py times at full size)."""
    from metalpathtracer_amd import host
    tmp = tmp_path_factory.mktemp("cfg4")
    _heightfield_obj(str(tmp / "hf.obj"), 501, seed=1)
    (tmp / "big.xml").write_text("""<Scene>
  <Mesh file="hf.obj" position="0,-10,-30" scale="1.0" albedo="0.7,0.7,0.75" emission="0,0,0" materialType="0" emissionPower="0"/>
  <Mesh file="hf.obj" position="0,35,-60" scale="0.6" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="-15,18,-10" radius="9" albedo="0.95,0.95,0.95" emission="0,0,0" materialType="-1" emissionPower="0"/>
  <Sphere position="15,18,-10" radius="9" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="0,60,-20" radius="10" albedo="0,0,0" emission="1,0.9,0.7" materialType="0" emissionPower="5"/>
</Scene>""")
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(str(tmp / "big.xml"), sc)
    assert st == 0, log
    assert sc.getPrimitiveCount() == 1000003
    sc.buildBVH(host.BVH_BINNED_CENTROID)
    return sc, sc.buffers()


def test_million_triangle_config_matches_the_oracle(gpu_ctx, million_triangle_scene):
    """configs[4] scene at 160x90 x 2 spp, depth 16, Scatter.h BSDFs: closest-first and reference-order pipelines vs
    the oracle, bit for bit."""
    from metalpathtracer_amd import capi, host
    sc, buf = million_triangle_scene
    gpu_ctx.upload_scene(*buf)
    info = gpu_ctx.accel_info()
    assert info["ordered_ok"] == 1 and info["nodes"] > info["lds_nodes"] > 16       # only the top of the tree fits LDS (30 KB per workgroup)
    W, H, spp = 160, 90, 2
    u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount())
    gpu_ctx.resize(W, H)
    gpu_ctx.set_uniforms(u)
    ref, ct = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buf, rng_mode=ob.RNG_PHILOX, bsdf_mode=ob.BSDF_SCATTER,
                        max_depth=16, accumulate=1, sample_count=spp, seed=(2, 7), threads=8)
    for pipe in (capi.PIPE_ORDERED, capi.PIPE_WAVELOCAL):
        gpu_ctx.clear_sum()
        gpu_ctx.reset_stats()
        gpu_ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=capi.BSDF_SCATTER, max_depth=16, sample_count=spp, seed=(2, 7),
                       pipeline=pipe)
        _same(gpu_ctx.read_sum(), ref)
        assert gpu_ctx.stats()["rays"] == ct["rays"]


@pytest.mark.parametrize("tree", ["reference", "device"])
def test_bunny20_full_size_properties(gpu_ctx, tree):
    """configs[2] at its full 1920x1080: determinism, sample-range and shard additivity, closest-first == reference
    order, and a band of rows against the oracle — on the reference's own tree and on the device-built one (leaves of <= 2, padded
    own boxes: what bench.py's bunny x20 workload and `--bvh auto` render; the oracle walks the tree read back from the device)."""
    from metalpathtracer_amd import capi
    W, H = 1920, 1080
    buf, uo = setup_tree(gpu_ctx, "bunny20.xml", W, H, tree)
    kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, seed=(1, 0))
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_count=4, **kw)
    a = gpu_ctx.read_sum()
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_count=4, **kw)
    _same(a, gpu_ctx.read_sum())                                                  # deterministic
    assert gpu_ctx.stats()["tree_parked"] > 0                                     # AUTO chose the closest-first pipeline
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_count=4, pipeline=capi.PIPE_WAVELOCAL, **kw)
    _same(a, gpu_ctx.read_sum())                                                  # closest-first == reference order
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_begin=0, sample_count=3, **kw)
    gpu_ctx.render(sample_begin=3, sample_count=1, **kw)
    _same(a, gpu_ctx.read_sum())                                                  # additive in samples
    total = np.zeros_like(a)
    for r in range(3):
        gpu_ctx.clear_sum()
        gpu_ctx.render(sample_count=4, shard_rank=r, shard_count=3, **kw)
        total += gpu_ctx.read_sum()
    _same(a, total)                                                               # additive in shards
    ref, _ = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=4, seed=(1, 0), rows=(560, 624))
    _same(a[560:624], ref[560:624])                                               # 64 rows through the bunnies vs the oracle


def test_smoke_entry_runs_the_default_pipeline():
    import __graft_entry__ as ge
    ge.smoke()


def _quat_act(angle, axis, v):
    """simd_act(simd::quatf(angle, axis), v) in float32, axis not normalised (what R/Renderer/Camera.h:53-60 calls)."""
    f = np.float32
    half = f(angle) * f(0.5)
    imag = (np.asarray(axis, f) * f(np.sin(half, dtype=f))).astype(f)
    real = f(np.cos(half, dtype=f))
    t = (np.cross(imag, v).astype(f) * f(2.0)).astype(f)
    return (np.asarray(v, f) + t * real + np.cross(imag, t).astype(f)).astype(f)


def _norm(v):
    v = np.asarray(v, np.float32)
    return (v / np.float32(np.sqrt(np.dot(v, v), dtype=np.float32))).astype(np.float32)


def test_camera_rotate_and_zoom_follow_the_reference_input_path(gpu_ctx):
    """mpt_renderer_input -> Camera::transformWithInputs (R/Renderer/Camera.h:49-89): rotate (two quaternion actions, the
    pitch axis NOT normalised) and zoom (clamped to [30, 120]) move the viewport exactly as a float32 restatement of the
    reference's lines does, reset the accumulation, and the frame drawn afterwards matches the oracle with those uniforms."""
    from conftest import oracle_scene, pixel_l2, scene_path
    from metalpathtracer_amd import host
    r = host.Renderer(0, scene_path("scene.xml"))
    W, H = 128, 72
    r.drawableSizeWillChange(W, H)
    r.draw()
    fwd, up, pos, fov = np.array([0, 0, -1], np.float32), np.array([0, 1, 0], np.float32), (0.0, 20.0, 50.0), np.float32(60.0)
    world_up = np.array([0, 1, 0], np.float32)
    for rot, zoom in (((40.0, -25.0), 0.0), ((-300.0, 120.0), 35.0), ((0.0, 0.0), -1000.0), ((15.0, 15.0), 2000.0)):
        r.input(rotate=rot, zoom=zoom)
        r.draw()
        if rot != (0.0, 0.0):                          # Camera.h:49-63
            right = np.cross(fwd, world_up).astype(np.float32)
            fwd = _norm(_quat_act(np.float32(-rot[1]) * np.float32(0.002), right, fwd))
            right = np.cross(fwd, world_up).astype(np.float32)
            up = _norm(np.cross(right, fwd).astype(np.float32))
            fwd = _norm(_quat_act(np.float32(-rot[0]) * np.float32(0.002), up, fwd))
        if zoom != 0.0:                                # Camera.h:65-72
            fov = np.float32(min(max(fov + np.float32(zoom) * np.float32(0.1), np.float32(30.0)), np.float32(120.0)))
        u = r.uniforms()
        assert u.frameCount == 0                       # a camera change resets the accumulation (Renderer.cpp:255-257)
        want = ob.Uniforms()
        ob.lib().orc_viewport((C.c_float * 3)(*pos), (C.c_float * 3)(*[float(x) for x in fwd]), (C.c_float * 3)(*[float(x) for x in up]),
                              C.c_float(float(fov)), C.c_float(W), C.c_float(H), C.byref(want))
        for name in ("viewportU", "viewportV", "firstPixelPosition", "cameraPosition"):
            assert list(getattr(u, name))[:3] == pytest.approx(list(getattr(want, name))[:3], abs=2e-6), (name, rot, zoom)
        got = r.readFrame()
        sc, buf = oracle_scene("scene.xml")
        ref, _ = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buf, rng_mode=ob.RNG_LITERAL, max_depth=32, accumulate=0, threads=8)
        assert pixel_l2(got, ref) < 1e-3
    assert fov == 120.0
    r.close()


def test_cli_camera_path_player_writes_frames(tmp_path):
    """mpt_render --camera-path: the scene.xml comments describe a camera that moves forward past the light sphere
    (R/scene.xml:6,10,14); 3 still frames, 4 steps forward with a mouse drag, a reset — one PPM per frame in --out-dir."""
    import json, subprocess
    from conftest import ROOT, scene_path
    path = tmp_path / "path.txt"
    path.write_text("# frames 0-2: still\n3\n4 w mouse 10 0   # forward, looking right\nscroll -50\nr\n")
    exe = os.path.join(ROOT, "metalpathtracer_amd", "lib", "mpt_render")
    out_dir = tmp_path / "runs"
    r = subprocess.run([exe, "--scene", scene_path("scene.xml"), "--width", "160", "--height", "90", "--rng", "literal",
                        "--camera-path", str(path), "--out-dir", str(out_dir)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    frames = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"frame"')]
    assert [f["frame"] for f in frames] == list(range(9))
    assert [f["frameCount"] for f in frames] == [1, 2, 3, 0, 0, 0, 0, 0, 0]        # every move / zoom / reset restarts
    assert frames[2]["camera"] == [0.0, 20.0, 50.0] and frames[6]["camera"][2] < 49.7 and frames[6]["forward"][0] > 0.05
    assert frames[7]["vfov"] == pytest.approx(65.0) and frames[8]["camera"] == [0.0, 20.0, 50.0] and frames[8]["vfov"] == 60.0
    files = sorted(os.listdir(out_dir))
    assert files == ["frame_%04d.ppm" % i for i in range(9)]
    assert all(os.path.getsize(out_dir / f) == len("P6\n160 90\n255\n") + 160 * 90 * 3 for f in files)


def test_c_abi_reduce_is_a_no_op_on_one_gpu(gpu_ctx):
    """mpt_comm_create_all / mpt_comm_create_rank / mpt_reduce_sum with N = 1 (librccl is not even opened): the HDR sum
    is untouched and in-flight renders are collected.  N > 1 needs as many GPUs: unmeasured on this one-GPU box."""
    from metalpathtracer_amd import capi
    buf, uo = setup(gpu_ctx, "scene.xml", 96, 54)
    gpu_ctx.clear_sum()
    gpu_ctx.render_async(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=2, seed=(1, 0))
    for comm in (capi.Comm.all([gpu_ctx]), capi.Comm.rank(gpu_ctx, 0, 1)):
        comm.reduce_sum(0)
        comm.close()
    got = gpu_ctx.read_sum()
    ref, _ = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=2, seed=(1, 0), threads=4)
    _same(got, ref)
    with pytest.raises(capi.MptError):
        capi.Comm.rank(gpu_ctx, 3, 2)


def test_closest_first_equals_reference_order_on_billions_of_rays(gpu_ctx):
    """Evidence at scale for the one step of the exactness argument that is not proven (sub-trees culled
    beyond best t * (1 + 2^-10) hold no hit in front of the winner): full-size renders — other seeds, sample ranges,
    depths, BSDF modes, tree builders than the headline test — through both pipelines, every float of the HDR sums
    compared.  The reference-order pipeline is itself checked against the oracle (test_gpu_parity.py), so equality here
    is equality with the oracle.  ~7 G rays, a few seconds of GPU time."""
    from metalpathtracer_amd import capi, host
    from conftest import scene_path
    total = 0
    cases = [("scene.xml", host.BVH_REFERENCE_SWEEP, 0, 8, 256, (7, 1), 1000), ("scene.xml", host.BVH_BINNED_CENTROID, 0, 32, 256, (9, 2), 0),
             ("scene.xml", host.BVH_GPU_LBVH, 0, 8, 256, (11, 3), 50000), ("glass.xml", host.BVH_REFERENCE_SWEEP, 1, 16, 256, (13, 4), 0),
             ("glass.xml", host.BVH_GPU_LBVH, 1, 32, 128, (15, 5), 7), ("bunny20.xml", host.BVH_REFERENCE_SWEEP, 0, 8, 128, (17, 6), 0),
             ("bunny20.xml", host.BVH_BINNED_CENTROID, 0, 8, 256, (19, 7), 300), ("bunny20.xml", host.BVH_GPU_LBVH, 1, 16, 128, (21, 8), 0)]
    for name, mode, bsdf, depth, spp, seed, sb in cases:
        sc = host.Scene()
        st, log = host.SceneLoader.LoadSceneFromXML(scene_path(name), sc)
        assert st == 0, log
        sc.buildBVH(mode)
        gpu_ctx.upload_scene(*sc.buffers())
        assert gpu_ctx.accel_info()["ordered_ok"] == 1
        W, H = 1920, 1080
        gpu_ctx.resize(W, H)
        gpu_ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
        img = {}
        for pipe in (capi.PIPE_WAVELOCAL, capi.PIPE_ORDERED):
            gpu_ctx.clear_sum()
            gpu_ctx.reset_stats()
            gpu_ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_begin=sb, sample_count=spp, seed=seed,
                           pipeline=pipe)
            img[pipe] = gpu_ctx.read_sum()
            rays = gpu_ctx.stats()["rays"]
        _same(img[capi.PIPE_WAVELOCAL], img[capi.PIPE_ORDERED])
        total += rays
    assert total > 5e9


@pytest.mark.parametrize("inplace", ["1", "40", "65"])
@pytest.mark.parametrize("name,bsdf", [("scene.xml", 0), ("glass.xml", 1), ("bunny20.xml", 0)])
def test_walking_in_place_or_parking_gives_the_same_image(name, bsdf, inplace, monkeypatch):
    """MPT_OT_INPLACE (read at mpt_create): a primary / ring-R step walks the tree at once when at least that many of its
    lanes need it (1: always, as soon as one lane does — partial waves of walkers beside lanes that are shaded; 65:
    never, every tree ray is parked first).  Same image as the oracle either way, with tiny walk budgets so that rays
    leave the in-place walk unfinished and are parked with their stacks."""
    from metalpathtracer_amd import capi
    monkeypatch.setenv("MPT_OT_INPLACE", inplace)
    monkeypatch.setenv("MPT_OT_BUDGETS", "2")
    ctx = capi.Context(0)
    try:
        W, H, spp, depth = 136, 77, 3, 12
        buf, uo = setup(ctx, name, W, H)
        ref, _ = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, accumulate=1, sample_count=spp,
                           seed=(4, 4), threads=8)
        ctx.clear_sum()
        ctx.reset_stats()
        ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_count=spp, seed=(4, 4), pipeline=capi.PIPE_ORDERED)
        _same(ctx.read_sum(), ref)
        st = ctx.stats()
        assert st["tree_parked"] > 0
    finally:
        ctx.close()


@pytest.mark.parametrize("occ", ["5", "6"])
@pytest.mark.parametrize("order", ["0", "1", "3"])
@pytest.mark.parametrize("name,W,H,bsdf,shards", [("bunny20.xml", 208, 117, 0, 1), ("glass.xml", 101, 67, 1, 3), ("scene.xml", 64, 40, 0, 5)])
def test_both_operating_points_and_every_tile_order_give_the_oracle_image(name, W, H, bsdf, shards, order, occ, monkeypatch):
    """Round 5: the closest-first kernel has two operating points (MPT_OT_OCC = 5: five workgroups of 256 threads per CU; 6: two of 768
    with a step's bookkeeping spilled around the walk loops) and the pass a tile order in which every claim range = XCD gets a vertical
    stripe of the image (MPT_TILE_ORDER = 3; 0 row-major with path -> pixel by arithmetic, 1 strided through the table).  Which waves
    trace which paths in which order never enters the arithmetic: every combination renders the oracle's image, every float — ragged
    sizes, tile shards whose tile counts do not divide by the eight claim ranges, fewer tiles than ranges (64 x 40 over 5 shards = 8
    tiles each) included — and the shards add up to it."""
    from metalpathtracer_amd import capi
    monkeypatch.setenv("MPT_OT_OCC", occ)
    monkeypatch.setenv("MPT_TILE_ORDER", order)
    ctx = capi.Context(0)
    try:
        spp, depth = 3, 10
        buf, uo = setup(ctx, name, W, H)
        ref, ct = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, accumulate=1, sample_count=spp, seed=(9, 2), threads=8)
        total = np.zeros_like(ref)
        rays = 0
        for r in range(shards):
            ctx.clear_sum()
            ctx.reset_stats()
            ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_count=spp, seed=(9, 2), pipeline=capi.PIPE_ORDERED,
                       shard_rank=r, shard_count=shards)
            total += ctx.read_sum()
            rays += ctx.stats()["rays"]
        _same(total, ref)
        assert rays == ct["rays"]
        ctx.clear_sum()                                    # the reference-order kernel walks the same tile tables
        ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=depth, sample_count=spp, seed=(9, 2), pipeline=capi.PIPE_WAVELOCAL)
        _same(ctx.read_sum(), ref)
    finally:
        ctx.close()
