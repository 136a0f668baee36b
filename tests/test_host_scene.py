"""Host scene layer (metalpathtracer_amd/csrc/host: Scene, SceneLoader, Camera/viewport) against the oracle's
restatement of R/Scene/Scene.h and R/Renderer/Renderer.cpp, plus structural properties of the BVH
(SURVEY.md 4 'property' row)."""
import ctypes as C

import numpy as np
import pytest

from conftest import CORNELL_CAM, host_scene, oracle_scene, scene_path
from metalpathtracer_amd import capi, host
from oracle import binding as ob


@pytest.mark.parametrize("name", ["scene.xml", "cornell.xml", "glass.xml", "bunny20.xml"])
def test_buffers_bit_equal_to_oracle(name):
    _, hb = host_scene(name)
    _, obuf = oracle_scene(name)
    for a, b in zip(hb, obuf):
        assert a.shape == b.shape
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def check_bvh(bvh, prims, idx, max_leaf):
    N, P = bvh.shape[0], prims.shape[0]
    left = bvh[:, 0, 3].view(np.int32)
    cnt = bvh[:, 1, 3].view(np.int32)
    seen_prims = np.zeros(P, np.int32)
    seen_nodes = np.zeros(N, np.int32)
    stack = [0]
    while stack:
        n = stack.pop()
        seen_nodes[n] += 1
        lo, hi = bvh[n, 0, :3], bvh[n, 1, :3]
        if cnt[n] > 0:
            assert cnt[n] <= max_leaf
            for k in range(left[n], left[n] + cnt[n]):
                p = idx[k]
                seen_prims[p] += 1
                if prims[p, 0, 3] == 0:
                    r = prims[p, 1, 0]
                    plo, phi = prims[p, 0, :3] - r, prims[p, 0, :3] + r
                else:
                    plo, phi = prims[p, :, :3].min(0), prims[p, :, :3].max(0)
                assert (plo >= lo).all() and (phi <= hi).all()
        else:
            l, r = left[n], -cnt[n]
            assert l == n + 1  # left child is emitted right after its parent (pre-order)
            for c in (l, r):
                assert 0 < c < N
                assert (bvh[c, 0, :3] >= lo).all() and (bvh[c, 1, :3] <= hi).all()
                stack.append(c)
    assert (seen_prims == 1).all()  # every primitive in exactly one leaf
    assert (seen_nodes == 1).all()  # a tree: every node reached once


@pytest.mark.parametrize("name", ["scene.xml", "cornell.xml"])
def test_reference_bvh_properties(name):
    _, (bvh, prims, mats, idx) = host_scene(name)
    check_bvh(bvh, prims, idx, max_leaf=8)


def test_binned_builder_properties():
    sc = host.Scene()
    st, _ = host.SceneLoader.LoadSceneFromXML(scene_path("scene.xml"), sc)
    assert st == 0
    sc.buildBVH(host.BVH_BINNED_CENTROID)
    bvh, prims, mats, idx = sc.buffers()
    check_bvh(bvh, prims, idx, max_leaf=8)
    assert (prims[:3, 0, 3] == 0).all()  # spheres still first


def test_defaults_and_sphere_first_ordering(tmp_path):
    (tmp_path / "t.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    xml = tmp_path / "s.xml"
    xml.write_text("""<Scene>
 <Mesh file="t.obj" position="1,2,3" scale="2" albedo="0.1,0.2,0.3" emission="0,0,0"/>
 <Sphere position="5,6,7" albedo="0.4,0.5,0.6" emission="0,0,0"/>
 <Sphere position="8,9,10" radius="3" albedo="0.7,0.8,0.9" emission="1,1,1" emissionPower="2" materialType="2"/>
</Scene>""")
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(str(xml), sc)
    assert st == 0 and "Loaded OBJ: 3 vertices, 1 triangles" in log
    assert (sc.getPrimitiveCount(), sc.getSphereCount(), sc.getTriangleCount()) == (3, 2, 1)
    sc.buildBVH()
    bvh, prims, mats, idx = sc.buffers()
    assert prims[:, 0, 3].tolist() == [0, 0, 1]                 # stable sphere-first (Scene.h:72-75)
    assert prims[0, 0, :3].tolist() == [5, 6, 7] and prims[0, 1, 0] == 1.0   # default radius 1
    assert prims[1, 1, 0] == 3.0 and mats[1, 0, 3] == 2.0 and mats[1, 1, 3] == 2.0
    assert mats[0, 0, 3] == 0.0 and mats[0, 1, 3] == 0.0        # default materialType / emissionPower 0
    np.testing.assert_array_equal(prims[2, :, :3], [[1, 2, 3], [3, 2, 3], [1, 4, 3]])  # pos + scale * v
    assert sc.getBVHNodeCount() == 1 and bvh[0, 1, 3].view(np.int32) == 3


def test_degenerate_inputs():
    sc = host.Scene()
    sc.buildBVH()
    assert sc.getBVHNodeCount() == 1 and sc.getPrimitiveCount() == 0   # empty scene: one empty leaf
    o = ob.OracleScene()
    o.build_bvh()
    assert o.node_count == 1
    # 40 identical zero-size primitives: parent area 0 -> single big leaf (Scene.h:231-232)
    for s_ in (sc, None):
        pass
    sc = host.Scene()
    o = ob.OracleScene()
    for _ in range(40):
        sc.addTriangle((1, 1, 1), (1, 1, 1), (1, 1, 1))
        o.add_triangle((1, 1, 1), (1, 1, 1), (1, 1, 1))
    sc.buildBVH()
    o.build_bvh()
    assert sc.getBVHNodeCount() == o.node_count == 1
    for a, b in zip(sc.buffers(), o.buffers()):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    # many primitives sharing the sort key on every axis but with extent: still identical to the oracle
    sc = host.Scene()
    o = ob.OracleScene()
    rng = np.random.default_rng(3)
    for i in range(300):
        a = rng.integers(0, 4, 3).astype(float)
        b = a + rng.random(3)
        c = a + rng.random(3)
        sc.addTriangle(a, b, c)
        o.add_triangle(a, b, c)
    sc.buildBVH()
    o.build_bvh()
    for a, b in zip(sc.buffers(), o.buffers()):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def test_viewport_and_seed_stream_match_oracle():
    for (W, H, cam) in ((1280, 720, None), (1920, 1080, None), (64, 64, CORNELL_CAM), (333, 77, CORNELL_CAM)):
        a = host.make_uniforms(W, H, 11, 7, cam=cam, random_seed=(0.1, 0.2, 0.3), frame_count=5)
        b = ob.make_uniforms(W, H, 11, 7, cam=cam, random_seed=(0.1, 0.2, 0.3), frame_count=5)
        assert bytes(a) == bytes(b)
    assert host.host_seed_sequence(6) == ob.host_seed_sequence(6)
    assert C.sizeof(capi.Uniforms) == 144
    assert host.camera_reset() == dict(pos=(0.0, 20.0, 50.0), fwd=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), vfov=60.0)


def test_image_writers(tmp_path):
    img = np.zeros((3, 5, 4), np.float32)
    img[0, 0] = [1, 0.5, 0.25, 1]
    img[2, 4] = [0.1, 0.2, 0.3, 1]
    assert host.write_pfm(str(tmp_path / "a.pfm"), img) == 0
    raw = (tmp_path / "a.pfm").read_bytes()
    assert raw.startswith(b"PF\n5 3\n-1.0\n")
    data = np.frombuffer(raw[len(b"PF\n5 3\n-1.0\n"):], np.float32).reshape(3, 5, 3)
    np.testing.assert_array_equal(data[::-1], img[..., :3])  # PFM rows go bottom-to-top
    assert host.write_ppm(str(tmp_path / "a.ppm"), img, gamma=1.0) == 0
    raw = (tmp_path / "a.ppm").read_bytes()
    assert raw.startswith(b"P6\n5 3\n255\n") and raw[len(b"P6\n5 3\n255\n"):][:3] == bytes([255, 128, 64])


def test_buffers_before_build_bvh_are_well_defined():
    """Scene::create*Buffer before buildBVH: no nodes, primitives in document order, identity index order (found by
    running the CPU suite under AddressSanitizer: the index copy used to read past a zero-length array)."""
    import numpy as np
    from conftest import ASSETS
    import os
    from metalpathtracer_amd import host
    sc = host.Scene()
    st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ASSETS, "cornell.xml"), sc)
    assert st == 0
    bvh, prims, mats, idx = sc.buffers()
    P = sc.getPrimitiveCount()
    assert bvh.shape[0] == 0 and prims.shape[0] == P and mats.shape[0] == P
    assert idx.tolist() == list(range(P))


@pytest.mark.parametrize("mode", [0, 1])
def test_parallel_bvh_build_is_identical_to_the_sequential_one(mode, monkeypatch):
    """Subtrees of big nodes are built concurrently and spliced in pre-order: the arrays must not depend on the
    number of threads (MPT_BUILD_THREADS=1 is the plain sequential recursion)."""
    import os
    import numpy as np
    from conftest import ASSETS
    from metalpathtracer_amd import host

    def build(threads):
        monkeypatch.setenv("MPT_BUILD_THREADS", str(threads))
        sc = host.Scene()
        st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ASSETS, "bunny20.xml"), sc)
        assert st == 0
        sc.buildBVH(mode)
        return sc.buffers()

    seq = build(1)
    for threads in (2, 8, 64):
        par = build(threads)
        for a, b in zip(seq, par):
            assert a.shape == b.shape
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_bvh_is_a_valid_partition_with_enclosing_boxes(mode, seed):
    """Structural invariants of both host builders on random scenes (spheres + triangles, clusters, duplicates):
    the leaves partition the primitive index list, every node box encloses its primitives, leaves hold <= 8."""
    import numpy as np
    from metalpathtracer_amd import host
    rng = np.random.default_rng(seed)
    sc = host.Scene()
    n_s, n_t = int(rng.integers(0, 6)), int(rng.integers(50, 700))
    for _ in range(n_s):
        sc.addSphere(tuple(rng.normal(0, 20, 3)), float(rng.uniform(0.5, 30)))
    centers = rng.normal(0, 15, (5, 3))
    for k in range(n_t):
        c = centers[rng.integers(0, 5)]
        v = c + rng.normal(0, 1.5, (3, 3))
        if k % 17 == 0:
            v[1] = v[0]                      # degenerate triangle
        if k % 29 == 0 and k:
            v = last                         # exact duplicate of an earlier triangle (equal sort keys)
        last = v
        sc.addTriangle(tuple(v[0]), tuple(v[1]), tuple(v[2]))
    sc.buildBVH(mode)
    bvh, prims, mats, idx = sc.buffers()
    P, N = sc.getPrimitiveCount(), sc.getBVHNodeCount()
    assert P == n_s + n_t and sorted(idx.tolist()) == list(range(P))
    lo = np.where(prims[:, 0, 3:4] == 1, prims[:, :, :3].min(1), prims[:, 0, :3] - prims[:, 1, 0:1])
    hi = np.where(prims[:, 0, 3:4] == 1, prims[:, :, :3].max(1), prims[:, 0, :3] + prims[:, 1, 0:1])
    covered = np.zeros(P, int)
    stack = [0]
    seen = 0
    while stack:
        n = stack.pop()
        seen += 1
        bmin, bmax = bvh[n, 0, :3], bvh[n, 1, :3]
        first, count = int(bvh[n, 0, 3].view(np.int32)), int(bvh[n, 1, 3].view(np.int32))
        if count > 0:                                         # leaf: [first, first + count) of the index list
            assert count <= 8
            members = idx[first:first + count]
            covered[first:first + count] += 1
            assert (lo[members] >= bmin - 0).all() and (hi[members] <= bmax + 0).all()
        else:                                                 # internal: children (first, -count)
            l, r = first, -count
            assert 0 < l < N and 0 < r < N and l != r
            for c in (l, r):
                assert (bvh[c, 0, :3] >= bmin).all() and (bvh[c, 1, :3] <= bmax).all()
                stack.append(c)
    assert seen == N and (covered == 1).all()
