"""GPU tests of the BVH built ON the GPU (mpt_build_bvh / Scene::buildBVH(GpuLbvh)): the arrays are a
well-formed tree in the reference's format with exactly nested boxes and leaves of <= 8 primitives, and — the tree is not
a parity target, the image is — the oracle renders from these arrays the very image the HIP pipelines render."""
import numpy as np
import pytest

from conftest import scene_path
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def capi_leaf(n):
    from metalpathtracer_amd import capi
    return capi.gpu_leaf_max(n)


def _check_tree(bvh, idx, prims):
    from metalpathtracer_amd import capi
    n_nodes, P = bvh.shape[0], prims.shape[0]
    leaf_max = capi.gpu_leaf_max(P)                                 # asked of the library (mpt_gpu_leaf_max), not re-derived here
    lf = bvh[:, 3].copy().view(np.int32)
    cnt = bvh[:, 7].copy().view(np.int32)
    assert sorted(idx.tolist()) == list(range(P))                  # every primitive in exactly one slot
    seen_nodes = np.zeros(n_nodes, bool)
    covered = np.zeros(P, np.int32)
    stack = [(0, None)]
    depth_max, leaves = 0, 0
    depth = {0: 1}
    while stack:
        n, parent = stack.pop()
        assert 0 <= n < n_nodes and not seen_nodes[n]
        seen_nodes[n] = True
        lo, hi = bvh[n, 0:3], bvh[n, 4:7]
        if parent is not None:                                     # nested exactly (the closest-first pipeline needs it)
            assert (lo >= bvh[parent, 0:3]).all() and (hi <= bvh[parent, 4:7]).all()
        depth_max = max(depth_max, depth[n])
        if cnt[n] > 0:
            leaves += 1
            assert cnt[n] <= leaf_max and 0 <= lf[n] and lf[n] + cnt[n] <= P
            covered[lf[n]:lf[n] + cnt[n]] += 1
            for k in range(cnt[n]):                                # the leaf box contains its primitives' boxes
                p = prims[idx[lf[n] + k]]
                if int(p[3]) == 0:
                    plo, phi = p[0:3] - p[4], p[0:3] + p[4]
                else:
                    v = np.stack([p[0:3], p[4:7], p[8:11]])
                    plo, phi = v.min(0), v.max(0)
                assert (plo >= lo).all() and (phi <= hi).all()
        else:
            for c in (lf[n], -cnt[n]):
                depth[c] = depth[n] + 1
                stack.append((c, n))
    assert seen_nodes.all() and (covered == 1).all()
    return leaves, depth_max


@pytest.mark.parametrize("name,bsdf", [("scene.xml", 0), ("glass.xml", 1), ("bunny20.xml", 0)])
def test_gpu_built_tree_is_well_formed_and_renders_the_oracle_image(gpu_ctx, name, bsdf):
    from metalpathtracer_amd import capi, host
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(scene_path(name), sc)
    assert st == 0, log
    sc.buildBVH(host.BVH_GPU_LBVH)
    bvh, prims, mats, idx = sc.buffers()
    bvh, prims = np.asarray(bvh).reshape(-1, 8), np.asarray(prims).reshape(-1, 12)
    leaves, depth = _check_tree(bvh, np.asarray(idx), prims)
    assert leaves * 8 >= prims.shape[0] and depth < 64              # leaves of <= 8; the reference's traversal stack holds 64 entries
    gpu_ctx.upload_scene(*sc.buffers())
    assert gpu_ctx.accel_info()["ordered_ok"] == 1
    W, H, spp = 160, 90, 3
    u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount())
    gpu_ctx.resize(W, H)
    gpu_ctx.set_uniforms(u)
    ref, ct = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), sc.buffers(), rng_mode=ob.RNG_PHILOX, bsdf_mode=bsdf, max_depth=8,
                        accumulate=1, sample_count=spp, seed=(4, 2), threads=8)
    for pipe in (capi.PIPE_ORDERED, capi.PIPE_WAVELOCAL):
        gpu_ctx.clear_sum()
        gpu_ctx.reset_stats()
        gpu_ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=8, sample_count=spp, seed=(4, 2), pipeline=pipe)
        np.testing.assert_array_equal(gpu_ctx.read_sum().view(np.uint32), ref.view(np.uint32))
        assert gpu_ctx.stats()["rays"] == ct["rays"]


@pytest.mark.parametrize("builder", ["sah", "ploc", "lbvh"])
def test_gpu_build_small_inputs_and_determinism(gpu_ctx, builder, monkeypatch):
    """1, 2, 3, 9 primitives (a single leaf; the smallest trees), duplicates (equal Morton codes, no separating plane), and the
    same arrays twice: the top-down builder numbers its nodes by atomics and must still write the same arrays every time."""
    # ("sah+refit" / "sah+sah": the own 4-wide tree from the builder's tree refitted, or from a second SAH over the leaves —
    #  by default the first when the scene has no sphere for the always list, the second otherwise, as in these four scenes)
    monkeypatch.setenv("MPT_GPU_BUILD", builder.split("+")[0])
    if "+" in builder:
        monkeypatch.setenv("MPT_OWN_TREE", builder.split("+")[1])
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 8, 9, 17, 100, 3000):
        prims = np.zeros((n, 12), np.float32)
        prims[:, 3] = 1.0
        v0 = rng.uniform(-5, 5, (n, 3))
        prims[:, 0:3], prims[:, 4:7], prims[:, 8:11] = v0, v0 + rng.uniform(0, 1, (n, 3)), v0 + rng.uniform(0, 1, (n, 3))
        if n >= 100:
            prims[n // 2:] = prims[:n // 2]                         # exact duplicates: ties in the sort keys
            prims[0, 3], prims[0, 4] = 0.0, 2.5                     # and one sphere
        bvh, idx, ms = gpu_ctx.build_bvh(prims)
        _check_tree(bvh, idx, prims)
        bvh2, idx2, _ = gpu_ctx.build_bvh(prims)
        np.testing.assert_array_equal(bvh.view(np.uint32), bvh2.view(np.uint32))
        np.testing.assert_array_equal(idx, idx2)
        assert (n <= capi_leaf(n)) == (bvh.shape[0] == 1)            # a scene that fits one leaf (mpt_gpu_leaf_max) is one node


def _digest(ctx):
    return ["%016x" % v for v in ctx.scene_digest()]


@pytest.mark.parametrize("builder", ["sah", "ploc", "lbvh"])
@pytest.mark.parametrize("name", ["scene.xml", "cornell.xml", "glass.xml", "bunny20.xml"])
def test_device_build_writes_the_recorded_arrays(gpu_ctx, name, builder, monkeypatch):
    """Every device array mpt_build_and_upload makes — threaded tree, primitive records, materials, own 4-wide tree, leaf boxes, always
    list, the reference-format tree — has the digest (mpt_scene_digest) recorded in tests/golden/devbuild_digests.json by the build of
    round 5's first half (tools/gpu_scene_digest.py --write, commit 192df6b): the builder's later rewrites — pooled outputs, level loops
    without read-backs, own radix sort, positions from the SAH, own tree by copy, picks up front, two streams — changed no word of any
    array.  A rebuild gives the same, and so does every fallback of the default builder: the bottom-up walk for the own tree, one
    stream, the material upload by the calling thread, and the 64-bit material sort that a collision of 32-bit keys falls back to (forced here by keeping 2 bits of the key)."""
    import json, os
    from conftest import ROOT
    from metalpathtracer_amd import host
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "devbuild_digests.json")))["%s/%s" % (name, builder)]
    monkeypatch.setenv("MPT_GPU_BUILD", builder)
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(scene_path(name), sc)
    assert st == 0, log
    prims, mats = sc.packed_primitives()
    names = "nodes prims mats own refleaf refbox always ref_bvh ref_idx n_nodes n_prims n_mats n_own n_leaves n_always depth".split()
    for k in range(2):
        gpu_ctx.build_and_upload(prims, mats)
        got = _digest(gpu_ctx)
        assert got == want, "build %d: other arrays: %s" % (k, [n for n, a, b in zip(names, got, want) if a != b])
    if builder == "sah":
        for env, val in (("MPT_OWN_TREE_WALK", "1"), ("MPT_BUILD_ONE_STREAM", "1"), ("MPT_BUILD_NO_HELPER", "1"), ("MPT_DEBUG_MAT_KEY_BITS", "2")):
            monkeypatch.setenv(env, val)
            gpu_ctx.build_and_upload(prims, mats)
            got = _digest(gpu_ctx)
            monkeypatch.delenv(env)
            assert got == want, "%s: other arrays: %s" % (env, [n for n, a, b in zip(names, got, want) if a != b])


def test_device_builds_of_two_contexts_at_once_and_of_alternating_sizes_write_the_recorded_arrays():
    """The builder's pooled state under the uses a host program makes of it.  (a) Two contexts of one GPU build in two threads at the same
    time, five times each — every build has its own scratch pool, pinned words, second stream and output block, and only the epoch
    counters of the level loops are shared: every build writes the recorded arrays.  (b) One context builds a large scene, a small one
    and the large one again: the output block of the scene before is the next build's when it is large enough (and the larger of the
    two is the one kept), a scene uploaded through mpt_upload_scene in between owns its arrays one by one — recorded arrays every time,
    and mpt_scene_digest of the uploaded scene reports no reference-format tree."""
    import json, os, threading
    from conftest import ROOT
    from metalpathtracer_amd import capi, host
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "devbuild_digests.json")))
    scenes = {}
    for name in ("bunny20.xml", "scene.xml"):
        sc = host.Scene()
        st, log = host.SceneLoader.LoadSceneFromXML(scene_path(name), sc)
        assert st == 0, log
        scenes[name] = (sc, sc.packed_primitives())
    os.environ.pop("MPT_GPU_BUILD", None)
    # (a)
    bad = []
    def work(name):
        try:
            ctx = capi.Context(0)
            for k in range(5):
                ctx.build_and_upload(*scenes[name][1])
                if _digest(ctx) != want[name + "/sah"]:
                    bad.append((name, k))
            ctx.close()
        except Exception as e:   # noqa: BLE001 — reported by the main thread
            bad.append((name, repr(e)))
    th = [threading.Thread(target=work, args=(n,)) for n in ("bunny20.xml", "scene.xml")]
    for t in th: t.start()
    for t in th: t.join()
    assert not bad, bad
    # (b)
    ctx = capi.Context(0)
    try:
        for name in ("bunny20.xml", "scene.xml", "bunny20.xml", "scene.xml", "scene.xml", "bunny20.xml"):
            ctx.build_and_upload(*scenes[name][1])
            assert _digest(ctx) == want[name + "/sah"], name
        sc = scenes["scene.xml"][0]
        sc.buildBVH()
        ctx.upload_scene(*sc.buffers())
        d = ctx.scene_digest()
        assert d[7] == 0 and d[8] == 0 and d[0] != 0 and d[10] == sc.getPrimitiveCount()
        ctx.build_and_upload(*scenes["bunny20.xml"][1])
        assert _digest(ctx) == want["bunny20.xml/sah"]
    finally:
        ctx.close()


@pytest.mark.parametrize("builder", ["sah", "sah+refit", "sah+sah", "ploc", "lbvh"])
@pytest.mark.parametrize("name,bsdf", [("scene.xml", 0), ("glass.xml", 1), ("bunny20.xml", 0), ("cornell.xml", 0)])
def test_build_and_upload_renders_the_oracle_image_of_its_own_tree(gpu_ctx, name, bsdf, builder, monkeypatch):
    """mpt_build_and_upload: build -> render without the host, with each of the three binary-tree builders (top-down binned
    SAH, the default; PLOC clustering; the plain Karras tree).  The tree it built comes back in the reference's format
    (mpt_download_bvh); it must be well formed, and the oracle must render from it exactly what both HIP pipelines render
    from the device-resident structures (threaded tree, own 4-wide tree, always list, materials: all derived on the device)."""
    from metalpathtracer_amd import capi, host
    monkeypatch.setenv("MPT_GPU_BUILD", builder)
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(scene_path(name), sc)
    assert st == 0, log
    sc.buildBVH()                                                  # (only to have the packed primitive / material arrays)
    _, prims, mats, _ = sc.buffers()
    ms = gpu_ctx.build_and_upload(prims, mats)
    assert 0.0 < ms < 200.0
    bvh, idx = gpu_ctx.download_bvh()
    p12 = np.asarray(prims).reshape(-1, 12)
    leaves, depth = _check_tree(bvh.reshape(-1, 8), idx, p12)
    info = gpu_ctx.accel_info()
    assert info["ordered_ok"] == 1 and info["reference_leaves"] == leaves and info["nodes"] >= 1
    assert info["always_spheres"] == int((p12[:, 3] == 0).sum())
    W, H, spp = 160, 90, 3
    from conftest import CORNELL_CAM
    u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), cam=CORNELL_CAM if name == "cornell.xml" else None)
    gpu_ctx.resize(W, H)
    gpu_ctx.set_uniforms(u)
    buffers = (bvh, prims, mats, idx)
    ref, ct = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buffers, rng_mode=ob.RNG_PHILOX, bsdf_mode=bsdf, max_depth=8,
                        accumulate=1, sample_count=spp, seed=(4, 2), threads=8)
    for pipe in (capi.PIPE_ORDERED, capi.PIPE_WAVELOCAL, capi.PIPE_MEGAKERNEL):
        gpu_ctx.clear_sum()
        gpu_ctx.reset_stats()
        gpu_ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=8, sample_count=spp, seed=(4, 2), pipeline=pipe,
                       flags=capi.FLAG_COUNT_WORK)
        np.testing.assert_array_equal(gpu_ctx.read_sum().view(np.uint32), ref.view(np.uint32))
        st = gpu_ctx.stats()
        assert st["rays"] == ct["rays"]
        if pipe != capi.PIPE_ORDERED:                               # the reference-order walk does the oracle's work, test for test
            assert (st["node_visits"], st["prim_tests"]) == (ct["node_pops"], ct["prim_tests"])
    # the host route over the same arrays (mpt_upload_scene of the downloaded tree) renders the same image
    gpu_ctx.upload_scene(*buffers)
    gpu_ctx.clear_sum()
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=8, sample_count=spp, seed=(4, 2))
    np.testing.assert_array_equal(gpu_ctx.read_sum().view(np.uint32), ref.view(np.uint32))


def test_build_and_upload_small_inputs_spheres_and_rays(gpu_ctx):
    """The smallest trees (1, 2, 3 primitives), spheres only, 17 spheres (no always list: reference-order pipelines only),
    duplicates; closest hits through both walks against the oracle on the downloaded tree."""
    from metalpathtracer_amd import capi
    rng = np.random.default_rng(11)
    for n, n_sph in ((1, 0), (1, 1), (2, 0), (3, 1), (9, 2), (40, 17), (300, 5)):
        prims = np.zeros((n, 12), np.float32)
        prims[:, 3] = 1.0
        v0 = rng.uniform(-5, 5, (n, 3))
        prims[:, 0:3], prims[:, 4:7], prims[:, 8:11] = v0, v0 + rng.uniform(-1, 1, (n, 3)), v0 + rng.uniform(-1, 1, (n, 3))
        prims[:n_sph, 3], prims[:n_sph, 4:12] = 0.0, 0.0
        prims[:n_sph, 4] = rng.uniform(0.3, 1.5, n_sph)
        if n == 300:
            prims[200:] = prims[100:200]                            # exact duplicates
        mats = np.zeros((n, 8), np.float32)
        mats[:, 0:3] = rng.choice([0.2, 0.5, 0.8], (n, 1))          # three distinct materials
        gpu_ctx.build_and_upload(prims, mats)
        bvh, idx = gpu_ctx.download_bvh()
        _check_tree(bvh.reshape(-1, 8), idx, prims)
        info = gpu_ctx.accel_info()
        assert info["ordered_ok"] == (1 if n_sph <= 16 else 0)
        m = 4096
        o = (rng.normal(size=(m, 3)) * 8).astype(np.float32)
        d = (rng.normal(size=(m, 3)) * 3 - o).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        t0, p0, n0, f0 = gpu_ctx.trace_rays(o, d)
        buffers = (bvh, prims.reshape(-1, 3, 4), mats.reshape(-1, 2, 4), idx)
        for i in range(0, m, 16):
            to, po, no, fo = ob.first_hit(o[i], d[i], buffers)
            assert po == p0[i] and (po < 0 or np.float32(to) == t0[i])
        if n_sph <= 16:
            t1, p1, n1, f1, fl = gpu_ctx.trace_rays_ordered(o, d)
            np.testing.assert_array_equal(t0.view(np.uint32), t1.view(np.uint32))
            np.testing.assert_array_equal(p0, p1)
        assert (p0 >= 0).sum() > 0 or n < 3


def test_build_and_upload_one_million_primitives_is_fast(gpu_ctx, tmp_path):
    """SURVEY 8 f-1 at config-4 size: 1,000,003 primitives ready to render in about ten milliseconds of wall time, the 80 MB
    upload included (measured 8.6-8.9 ms, 4.5-4.8 of them on the device; the reference's builder: 8.2 s on one core;
    mpt_build_bvh + mpt_upload_scene: 0.55 s), and the render agrees with the reference-order one."""
    import time
    from metalpathtracer_amd import capi, host
    from test_gpu_parity import _heightfield_obj
    _heightfield_obj(str(tmp_path / "hf.obj"), 501, seed=1)
    (tmp_path / "big.xml").write_text("""<Scene>
  <Mesh file="hf.obj" position="0,-10,-30" scale="1.0" albedo="0.7,0.7,0.75" emission="0,0,0" materialType="0" emissionPower="0"/>
  <Mesh file="hf.obj" position="0,35,-60" scale="0.6" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="-15,18,-10" radius="9" albedo="0.95,0.95,0.95" emission="0,0,0" materialType="-1" emissionPower="0"/>
  <Sphere position="15,18,-10" radius="9" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="0,60,-20" radius="10" albedo="0,0,0" emission="1,0.9,0.7" materialType="0" emissionPower="5"/>
</Scene>""")
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(str(tmp_path / "big.xml"), sc)
    assert st == 0 and sc.getPrimitiveCount() == 1000003, log
    prims, mats = sc.packed_primitives()
    gpu_ctx.build_and_upload(prims, mats)                          # warm-up (allocator, code objects)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        ms = gpu_ctx.build_and_upload(prims, mats)
        best = min(best, time.perf_counter() - t0)
    info = gpu_ctx.accel_info()
    assert info["ordered_ok"] == 1 and info["always_spheres"] == 3 and info["nodes"] > 100000
    assert best < 0.020 and ms < 12.0, "1 M primitives took %.1f ms of wall time (device %.1f ms)" % (best * 1e3, ms)
    W, H = 320, 180
    gpu_ctx.resize(W, H)
    gpu_ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
    img = {}
    for pipe in (capi.PIPE_ORDERED, capi.PIPE_WAVELOCAL):
        gpu_ctx.clear_sum()
        gpu_ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=capi.BSDF_SCATTER, max_depth=16, sample_count=2, seed=(2, 7), pipeline=pipe)
        img[pipe] = gpu_ctx.read_sum()
    np.testing.assert_array_equal(img[capi.PIPE_ORDERED].view(np.uint32), img[capi.PIPE_WAVELOCAL].view(np.uint32))


def test_cli_renders_bunny20_on_the_device_built_tree(gpu_ctx, tmp_path):
    """mpt_render (Renderer facade): a batch render of a scene of >= 8192 primitives builds its tree on the device (--bvh auto,
    the default) and gives the image mpt_build_and_upload + mpt_render give through the C ABI."""
    import json, os, subprocess
    from conftest import ROOT
    from metalpathtracer_amd import capi, host
    exe = os.path.join(ROOT, "metalpathtracer_amd", "lib", "mpt_render")
    out = str(tmp_path / "o.pfm")
    r = subprocess.run([exe, "--scene", scene_path("bunny20.xml"), "--width", "96", "--height", "54", "--spp", "2", "--depth", "8", "--seed", "5",
                        "--out", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "build -> render without the host" in r.stdout and "BVH built on the device" in r.stdout
    raw = open(out, "rb").read()
    hdr = b"PF\n96 54\n-1.0\n"
    img = np.frombuffer(raw[len(hdr):], np.float32).reshape(54, 96, 3)[::-1]
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(scene_path("bunny20.xml"), sc)
    assert st == 0, log
    sc.buildBVH()
    prims, mats = sc.packed_primitives()
    gpu_ctx.build_and_upload(prims, mats)
    gpu_ctx.resize(96, 54)
    gpu_ctx.set_uniforms(host.make_uniforms(96, 54, sc.getPrimitiveCount(), sc.getTriangleCount()))
    gpu_ctx.clear_sum()
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=2, seed=(5, 0))
    np.testing.assert_array_equal(img, gpu_ctx.read_sum()[..., :3] * np.float32(0.5))


def test_build_and_upload_survives_degenerate_input(gpu_ctx):
    """Primitives the builders must not choke on: NaN and infinite vertices, zero-area triangles, every primitive identical
    (all Morton codes equal), a zero-radius sphere.  The build terminates, the tree is well formed where boxes are finite, and
    the reference-order walk on the device equals the oracle on the downloaded tree."""
    rng = np.random.default_rng(21)
    n = 257
    prims = np.zeros((n, 12), np.float32)
    prims[:, 3] = 1.0
    v0 = rng.uniform(-4, 4, (n, 3))
    prims[:, 0:3], prims[:, 4:7], prims[:, 8:11] = v0, v0 + rng.uniform(-1, 1, (n, 3)), v0 + rng.uniform(-1, 1, (n, 3))
    prims[5, 4] = np.nan
    prims[6, 0:3] = np.inf
    prims[7, 8] = -np.inf
    prims[8, 4:7] = prims[8, 0:3]                                  # zero area
    prims[9, 4:7] = prims[9, 8:11] = prims[9, 0:3]                # a point
    prims[0, 3], prims[0, 4:12] = 0.0, 0.0                          # a sphere of radius 0
    mats = np.zeros((n, 8), np.float32)
    mats[:, 0:3] = 0.5
    for variant in ("mixed", "identical"):
        p = prims.copy()
        if variant == "identical":
            p[:] = prims[20]
        gpu_ctx.build_and_upload(p, mats)
        bvh, idx = gpu_ctx.download_bvh()
        assert sorted(idx.tolist()) == list(range(n)) and 1 <= bvh.shape[0] <= 2 * n - 1
        m = 2048
        o = (rng.normal(size=(m, 3)) * 7).astype(np.float32)
        d = (rng.normal(size=(m, 3)) * 2 - o).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        t0, p0, n0, f0 = gpu_ctx.trace_rays(o, d)
        buffers = (bvh, p.reshape(-1, 3, 4), mats.reshape(-1, 2, 4), idx)
        for i in range(0, m, 8):
            to, po, no, fo = ob.first_hit(o[i], d[i], buffers)
            assert po == p0[i] and (po < 0 or np.float32(to) == t0[i]), (variant, i)
        # (how much is hit is the reference's business: a NaN vertex poisons the boxes above it, for the oracle and the device alike)


def _sah_cost(bvh, idx, prims):
    """SAH cost of the part of a tree (reference buffer format) that holds triangles only: sum over those nodes of box area
    over the area of the triangles' bounding box, times 1 for an inner node and the primitive count for a leaf — the quantity
    the builders minimise.  (Nodes with a sphere below them are left out: scene.xml's ground sphere of radius 1e4 would be
    all there is to see otherwise.)"""
    raw = np.ascontiguousarray(np.asarray(bvh, np.float32).reshape(-1, 8))
    lf, cnt = raw[:, 3].copy().view(np.int32), raw[:, 7].copy().view(np.int32)
    P = np.asarray(prims, np.float32).reshape(-1, 12)
    idx = np.asarray(idx)
    tri = P[:, 3] == 1.0
    v = np.stack([P[tri, 0:3], P[tri, 4:7], P[tri, 8:11]], 1)
    e = v.max((0, 1)).astype(np.float64) - v.min((0, 1)).astype(np.float64)
    norm = e[0] * e[1] + e[1] * e[2] + e[2] * e[0]
    ext = np.maximum(raw[:, 4:7].astype(np.float64) - raw[:, 0:3].astype(np.float64), 0.0)
    area = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
    has_sphere = np.zeros(len(raw), bool)
    order, stack = [], [0]
    while stack:
        n = stack.pop()
        order.append(n)
        if cnt[n] <= 0:
            stack += [int(lf[n]), int(-cnt[n])]
    for n in reversed(order):
        if cnt[n] > 0:
            has_sphere[n] = not tri[idx[lf[n]:lf[n] + cnt[n]]].all()
        else:
            has_sphere[n] = has_sphere[lf[n]] or has_sphere[-cnt[n]]
    w = np.where(cnt > 0, cnt, 1).astype(np.float64)
    keep = ~has_sphere
    return float((area[keep] * w[keep]).sum() / norm)


@pytest.mark.parametrize("name", ["bunny20.xml", "scene.xml"])
def test_device_sah_tree_is_as_good_as_the_host_binned_builders(gpu_ctx, name, monkeypatch):
    """The device builder runs the host binned builder's algorithm (over the triangles; the spheres hang under its root): the SAH
    cost of the triangle part of its tree must match the host's within 3 %, and beat the clustering builders', at the same
    leaf size."""
    from metalpathtracer_amd import host
    monkeypatch.setenv("MPT_LBVH_LEAF", "2")
    monkeypatch.setenv("MPT_BINNED_LEAF", "2")
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(scene_path(name), sc)
    assert st == 0, log
    sc.buildBVH(host.BVH_BINNED_CENTROID)
    b = sc.buffers()
    ref = _sah_cost(b[0], b[3], b[1])
    cost = {}
    for builder in ("sah", "ploc", "lbvh"):
        monkeypatch.setenv("MPT_GPU_BUILD", builder)
        tree, prims, _, idx = host.make_ready(gpu_ctx, sc, host.BVH_DEVICE)
        cost[builder] = _sah_cost(tree, idx, prims)
    assert cost["sah"] < 1.03 * ref, (cost, ref)
    assert cost["sah"] < cost["ploc"] and cost["sah"] < cost["lbvh"], (cost, ref)


def test_flat_walls_stay_visible_on_the_products_trees(gpu_ctx, monkeypatch):
    """The reference's slab test never enters a box of zero thickness (PathTracing.h:68), so a leaf that holds only the two
    coplanar halves of an axis-aligned Cornell-box wall would make the wall invisible.  The reference's own leaves (<= 8,
    mixed orientations) are not flat in cornell.xml; the product's builders make small leaves and therefore give every
    primitive box a thickness: their trees must render what the reference's tree renders, up to the rays whose answer depends
    on the visit order."""
    from conftest import CORNELL_CAM, pixel_l2
    from metalpathtracer_amd import capi, host
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(scene_path("cornell.xml"), sc)
    assert st == 0, log
    W, H, spp = 192, 192, 8
    gpu_ctx.resize(W, H)
    gpu_ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), cam=CORNELL_CAM))

    def render(bvh):
        host.make_ready(gpu_ctx, sc, bvh)
        gpu_ctx.clear_sum()
        gpu_ctx.reset_stats()
        gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=spp, seed=(3, 1))
        return gpu_ctx.read_sum() / spp, gpu_ctx.stats()["rays"]

    ref, rays = render(host.BVH_REFERENCE_SWEEP)
    assert rays > 2.3 * W * H * spp                                # the box is closed on five sides: paths bounce
    cases = [("binned", host.BVH_BINNED_CENTROID, None), ("gpu through the host", host.BVH_GPU_LBVH, "sah")]
    cases += [("device " + b, host.BVH_DEVICE, b) for b in ("sah", "ploc", "lbvh")]
    for leaf in ("2", "6"):
        monkeypatch.setenv("MPT_LBVH_LEAF", leaf)
        monkeypatch.setenv("MPT_BINNED_LEAF", leaf)
        for tag, bvh, builder in cases:
            if builder:
                monkeypatch.setenv("MPT_GPU_BUILD", builder)
            img, r = render(bvh)
            assert abs(r - rays) <= 1e-3 * rays, (tag, leaf, r, rays)
            assert pixel_l2(img, ref) < 1e-3, (tag, leaf)


@pytest.mark.parametrize("name,bsdf,W,H,spp", [("scene.xml", 0, 240, 135, 4), ("glass.xml", 1, 240, 135, 4), ("bunny20.xml", 0, 160, 90, 2)])
def test_throughput_trees_render_the_image_of_the_reference_tree(gpu_ctx, name, bsdf, W, H, spp):
    """`--bvh auto` / bench.py build the tree on the device; the drop-in default is the reference's own builder.  Both must show
    the same picture: equal up to the rays whose answer depends on the visit order (ties between primitives), i.e. nearly
    the same ray count and a per-pixel L2 far below the north-star tolerance."""
    from conftest import pixel_l2
    from metalpathtracer_amd import capi, host
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(scene_path(name), sc)
    assert st == 0, log
    gpu_ctx.resize(W, H)
    gpu_ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
    out = {}
    for bvh in (host.BVH_REFERENCE_SWEEP, host.BVH_BINNED_CENTROID, host.BVH_DEVICE):
        host.make_ready(gpu_ctx, sc, bvh)
        gpu_ctx.clear_sum()
        gpu_ctx.reset_stats()
        gpu_ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=8, sample_count=spp, seed=(9, 4))
        out[bvh] = (gpu_ctx.read_sum() / spp, gpu_ctx.stats()["rays"])
    ref, rays = out[host.BVH_REFERENCE_SWEEP]
    for bvh in (host.BVH_BINNED_CENTROID, host.BVH_DEVICE):
        img, r = out[bvh]
        assert abs(r - rays) <= 1e-4 * rays + 4, (bvh, r, rays)
        assert pixel_l2(img, ref) < 1e-4, bvh


@pytest.mark.parametrize("builder", ["sah", "ploc"])
def test_builders_at_the_kernels_size_boundaries(gpu_ctx, builder, monkeypatch):
    """Primitive counts on both sides of every switch inside the builders — the lane-per-item kernel (<= 8 items), the
    workgroup-chunk kernels (>= 2048), the full-binning limit and the leaf-size / pipeline switch (8192) — with 0, 1, 16 and 17
    spheres (hoisted under the root up to 16): well-formed tree, the same arrays on a second build, and both walks agree with
    the oracle's closest hit on the downloaded tree."""
    monkeypatch.setenv("MPT_GPU_BUILD", builder)
    rng = np.random.default_rng(23)
    cases = [(5, 0), (8, 1), (9, 0), (17, 16), (33, 17), (257, 1), (2047, 0), (2048, 3), (2049, 16), (4100, 1), (8191, 0), (8192, 2), (8300, 17)]
    for n, n_sph in cases:
        prims = np.zeros((n, 12), np.float32)
        prims[:, 3] = 1.0
        v0 = rng.uniform(-20, 20, (n, 3))
        prims[:, 0:3], prims[:, 4:7], prims[:, 8:11] = v0, v0 + rng.uniform(-1, 1, (n, 3)), v0 + rng.uniform(-1, 1, (n, 3))
        prims[:n_sph, 3], prims[:n_sph, 4:12] = 0.0, 0.0
        prims[:n_sph, 4] = rng.uniform(0.5, 3.0, n_sph)
        if n_sph:
            prims[0, 0:3], prims[0, 4] = (0.0, -1000.0, 0.0), 980.0     # a ground sphere far larger than the rest
        mats = np.zeros((n, 8), np.float32)
        mats[:, 0:3] = 0.5
        gpu_ctx.build_and_upload(prims, mats)
        bvh, idx = gpu_ctx.download_bvh()
        _check_tree(bvh.reshape(-1, 8), idx, prims)
        gpu_ctx.build_and_upload(prims, mats)
        bvh2, idx2 = gpu_ctx.download_bvh()
        np.testing.assert_array_equal(bvh.view(np.uint32), bvh2.view(np.uint32))
        np.testing.assert_array_equal(idx, idx2)
        m = 2048
        o = (rng.normal(size=(m, 3)) * 25).astype(np.float32)
        d = (rng.normal(size=(m, 3)) * 10 - o).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        t0, p0, n0, f0 = gpu_ctx.trace_rays(o, d)
        buffers = (bvh, prims.reshape(-1, 3, 4), mats.reshape(-1, 2, 4), idx)
        for i in range(0, m, 64):
            to, po, no, fo = ob.first_hit(o[i], d[i], buffers)
            assert po == p0[i] and (po < 0 or np.float32(to) == t0[i]), (n, n_sph, i)
        if n_sph <= 16:
            assert gpu_ctx.accel_info()["ordered_ok"] == 1
            t1, p1, n1, f1, fl = gpu_ctx.trace_rays_ordered(o, d)
            np.testing.assert_array_equal(t0.view(np.uint32), t1.view(np.uint32))
            np.testing.assert_array_equal(p0, p1)
        assert n < 257 or (p0 >= 0).sum() > m // 50, (n, n_sph)


def test_config4_at_its_real_size_on_the_device_built_tree(gpu_ctx, tmp_path):
    """BASELINE.json configs[4] at its real size on the route `--bvh auto` takes: the 1,000,003-primitive scene (two 500 K-triangle
    height fields, one of them glass; mirror and glass spheres; an emitter) built ON THE DEVICE, 1920x1080, one GPU's 1/8 tile
    shard, 4096 spp, depth 16, Scatter.h BSDFs = 1.06 G paths in one pass: deterministic, and the closest-first pipeline
    (MPT_PIPE_AUTO for this scene) gives bit for bit what the reference-order pipeline gives; then, at 4 spp, a 32-row band
    of the unsharded image against the oracle walking the tree that mpt_download_bvh returns."""
    import time
    from metalpathtracer_amd import capi, host
    from test_gpu_parity import _heightfield_obj
    _heightfield_obj(str(tmp_path / "hf.obj"), 501, seed=1)
    (tmp_path / "big.xml").write_text("""<Scene>
  <Mesh file="hf.obj" position="0,-10,-30" scale="1.0" albedo="0.7,0.7,0.75" emission="0,0,0" materialType="0" emissionPower="0"/>
  <Mesh file="hf.obj" position="0,35,-60" scale="0.6" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="-15,18,-10" radius="9" albedo="0.95,0.95,0.95" emission="0,0,0" materialType="-1" emissionPower="0"/>
  <Sphere position="15,18,-10" radius="9" albedo="1,1,1" emission="0,0,0" materialType="1.5" emissionPower="0"/>
  <Sphere position="0,60,-20" radius="10" albedo="0,0,0" emission="1,0.9,0.7" materialType="0" emissionPower="5"/>
</Scene>""")
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(str(tmp_path / "big.xml"), sc)
    assert st == 0 and sc.getPrimitiveCount() == 1000003, log
    buf = host.make_ready(gpu_ctx, sc, host.BVH_DEVICE)
    info = gpu_ctx.accel_info()
    assert info["ordered_ok"] == 1 and info["auto_pipeline"] == capi.PIPE_ORDERED and gpu_ctx.build_info()["built_leaf_max"] == 2
    W, H = 1920, 1080
    u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount())
    gpu_ctx.resize(W, H)
    gpu_ctx.set_uniforms(u)
    kw = dict(rng_mode=capi.RNG_PHILOX, bsdf_mode=capi.BSDF_SCATTER, max_depth=16, seed=(1, 0))
    shard = dict(shard_rank=3, shard_count=8)
    img, ms = {}, {}
    for tag, pipe in (("auto", capi.PIPE_AUTO), ("again", capi.PIPE_AUTO), ("reference order", capi.PIPE_WAVELOCAL)):
        gpu_ctx.clear_sum()
        gpu_ctx.reset_stats()
        gpu_ctx.render(sample_count=4096, pipeline=pipe, **shard, **kw)
        s = gpu_ctx.stats()
        assert s["trace_launches"] == 1 and s["paths"] == 4050 * 64 * 4096        # 240 x 135 tiles / 8 ranks: ONE pass of 1.06 G paths
        assert (s["tree_parked"] > 0) == (pipe == capi.PIPE_AUTO)
        img[tag], ms[tag] = gpu_ctx.read_sum(), (s["total_ms"], s["rays"])
    print("config 4 shard: k_ordered %.1f ms (%.2f Grays/s), k_wavelocal %.1f ms, %d rays" % (
        ms["auto"][0], ms["auto"][1] / ms["auto"][0] / 1e6, ms["reference order"][0], ms["auto"][1]))
    assert np.isfinite(img["auto"]).all() and img["auto"][..., 3].max() > 0
    np.testing.assert_array_equal(img["auto"].view(np.uint32), img["again"].view(np.uint32))              # deterministic
    np.testing.assert_array_equal(img["auto"].view(np.uint32), img["reference order"].view(np.uint32))    # closest-first == reference order
    assert ms["auto"][1] == ms["reference order"][1]
    owned = img["auto"][..., 3] > 0
    ty, tx = np.nonzero(owned)
    assert set(((ty // 8) * 240 + tx // 8) % 8) == {3}                                # only rank 3's tiles were touched
    del img
    # ---- a band of rows against the oracle on the downloaded tree (4 spp, unsharded) ------------------------------------------
    rows = (600, 632)                                                                 # through the lower height field
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_count=4, **kw)
    whole = gpu_ctx.read_sum()
    ref, _ = ob.render(ob.Uniforms.from_buffer_copy(bytes(u)), buf, rng_mode=ob.RNG_PHILOX, bsdf_mode=ob.BSDF_SCATTER, max_depth=16,
                       accumulate=1, sample_count=4, seed=(1, 0), rows=rows, threads=8)
    np.testing.assert_array_equal(whole[rows[0]:rows[1]].view(np.uint32), ref[rows[0]:rows[1]].view(np.uint32))


def test_a_ground_quad_hangs_under_the_root_like_a_ground_sphere(gpu_ctx):
    """Round 4: the device builder keeps every item whose box dwarfs the rest out of its SAH (>= 2^10 x the median extent by
    binades), not only spheres: a ground QUAD of two 2 x 10^4-unit triangles under the bunny stretches the root box — and the
    bounds of the box centres the SAH bins over — exactly as scene.xml's ground sphere does.  The two triangles must hang under
    the root in a leaf of their own, the root's other child must be the SAH tree of the mesh with the mesh's own box, and that
    tree must cost what the tree of the mesh ALONE costs (the quad changes nothing below the root); both walks agree with the
    oracle on the downloaded tree."""
    from metalpathtracer_amd import capi, host
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(scene_path("scene.xml"), sc)
    assert st == 0, log
    sc.sortPrimitives()
    prims0, mats0 = sc.packed_primitives()
    tri = np.asarray(prims0).reshape(-1, 12)
    tri = tri[tri[:, 3] == 1.0]                                       # the bunny alone (the three spheres dropped)
    mats = np.zeros((len(tri) + 2, 8), np.float32)
    mats[:, 0:3] = 0.7
    S = 1.0e4
    quad = np.zeros((2, 12), np.float32)
    quad[:, 3] = 1.0
    quad[0, 0:3], quad[0, 4:7], quad[0, 8:11] = (-S, -0.5, -S), (S, -0.5, -S), (S, -0.5, S)
    quad[1, 0:3], quad[1, 4:7], quad[1, 8:11] = (-S, -0.5, -S), (S, -0.5, S), (-S, -0.5, S)
    alone = tri.copy()
    gpu_ctx.build_and_upload(alone, mats[:len(alone)])
    bvh_a, idx_a = gpu_ctx.download_bvh()
    cost_alone = _sah_cost(bvh_a, idx_a, alone)
    both = np.concatenate([tri[:100], quad, tri[100:]])               # (the quad somewhere in the middle of the array)
    gpu_ctx.build_and_upload(both, mats)
    bvh, idx = gpu_ctx.download_bvh()
    raw = bvh.reshape(-1, 8)
    _check_tree(raw, idx, both)
    lf, cnt = raw[:, 3].copy().view(np.int32), raw[:, 7].copy().view(np.int32)
    assert cnt[0] <= 0                                                # the root is an inner node ...
    kids = [int(lf[0]), int(-cnt[0])]
    leaf = [k for k in kids if cnt[k] > 0]
    assert len(leaf) == 1 and cnt[leaf[0]] == 2                       # ... one child is a leaf of two primitives:
    assert sorted(idx[lf[leaf[0]]:lf[leaf[0]] + 2].tolist()) == [100, 101]   # the quad
    other = [k for k in kids if k != leaf[0]][0]
    v = np.stack([tri[:, 0:3], tri[:, 4:7], tri[:, 8:11]], 1)
    lo, hi = v.min((0, 1)), v.max((0, 1))
    pad = 1e-3 * (hi - lo).max()
    assert (raw[other, 0:3] >= lo - pad).all() and (raw[other, 4:7] <= hi + pad).all()   # the mesh's own box: not stretched by the quad
    # cost of the mesh part (nodes without a quad triangle below them), by the measure of _sah_cost with the quad masked out like a sphere
    masked = both.copy()
    masked[100:102, 3] = 0.0
    cost_with = _sah_cost(raw, idx, masked)
    assert abs(cost_with - cost_alone) <= 0.03 * cost_alone, (cost_with, cost_alone)
    rng = np.random.default_rng(5)
    m = 4096
    o = (rng.normal(size=(m, 3)) * 30 + (0, 20, 0)).astype(np.float32)
    d = (rng.normal(size=(m, 3)) * 8 + (0, 5, 0) - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t0, p0, _, _ = gpu_ctx.trace_rays(o, d)
    t1, p1, _, _, _ = gpu_ctx.trace_rays_ordered(o, d)
    np.testing.assert_array_equal(t0.view(np.uint32), t1.view(np.uint32))
    np.testing.assert_array_equal(p0, p1)
    buffers = (bvh, both.reshape(-1, 3, 4), mats.reshape(-1, 2, 4), idx)
    for i in range(0, m, 97):
        to, po, _, _ = ob.first_hit(o[i], d[i], buffers)
        assert po == p0[i] and (po < 0 or np.float32(to) == t0[i]), i
    assert (p0 >= 0).sum() > m // 4
