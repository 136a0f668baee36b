"""Committed golden fixtures (tests/golden/*.npy, made by tests/golden/make_golden.py with the oracle).

CPU half: the oracle still reproduces them bit for bit (guards the oracle itself).
GPU half (-m gpu): the HIP path, fed by the product's own host layer, reproduces them — bit-exact for the
philox cases (integer RNG + IEEE FP32 in the same operation order on both sides), and within the
north-star tolerance (per-pixel L2 < 1e-3) for the literal cases, whose cos/sin come from libm on the CPU
and from the device math library on the GPU."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, host_scene, oracle_scene, pixel_l2
from oracle import binding as ob

MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))
L2_TOL = 1e-3  # BASELINE.json north_star: per-pixel L2 error < 1e-3


def _uniform_kwargs(m):
    uk = dict(m["uniforms"])
    if uk.get("random_seed") == "host":
        uk["random_seed"] = ob.host_seed_sequence(3)
    return uk


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_oracle_reproduces_golden(name):
    m = MANIFEST[name]
    sc, buf = oracle_scene(m["scene"])
    assert sc.prim_count == m["prims"] and sc.node_count == m["nodes"]
    u = ob.make_uniforms(m["width"], m["height"], sc.prim_count, sc.triangle_count, cam=m["camera"],
                         **_uniform_kwargs(m))
    rk = dict(m["render"])
    if "seed" in rk:
        rk["seed"] = tuple(rk["seed"])
    img, ct = ob.render(u, buf, threads=4, **rk)
    want = np.load(os.path.join(GOLDEN, name + ".npy"))
    np.testing.assert_array_equal(img.view(np.uint32), want.view(np.uint32))
    assert "%016x" % ob.fnv1a64(img) == m["fnv1a64"]
    assert ct == m["counters"]


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1, 2])
@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_hip_reproduces_golden(gpu_ctx, name, pipeline):
    from metalpathtracer_amd import capi, host
    m = MANIFEST[name]
    sc, buf = host_scene(m["scene"])
    gpu_ctx.upload_scene(*buf)
    u = host.make_uniforms(m["width"], m["height"], sc.getPrimitiveCount(), sc.getTriangleCount(), cam=m["camera"],
                           **_uniform_kwargs(m))
    gpu_ctx.resize(m["width"], m["height"])
    gpu_ctx.set_uniforms(u)
    rk = m["render"]
    want = np.load(os.path.join(GOLDEN, name + ".npy"))
    kw = dict(rng_mode=rk["rng_mode"], bsdf_mode=rk.get("bsdf_mode", 0), max_depth=rk["max_depth"], pipeline=pipeline,
              flags=capi.FLAG_COUNT_WORK)
    gpu_ctx.reset_stats()
    if rk["accumulate"] == 0:
        gpu_ctx.draw(**kw)
        got = gpu_ctx.read_frame()
        assert pixel_l2(got, want) < L2_TOL
        assert np.abs(got - want).max() < 1e-5   # observed: a few ulp
    else:
        gpu_ctx.clear_sum()
        gpu_ctx.render(sample_count=rk["sample_count"], seed=tuple(rk["seed"]), **kw)
        got = gpu_ctx.read_sum()
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
        st = gpu_ctx.stats()
        c = m["counters"]
        assert (st["paths"], st["rays"], st["node_visits"], st["aabb_hits"], st["prim_tests"]) == (
            c["paths"], c["rays"], c["node_pops"], c["aabb_pass"], c["prim_tests"])


def test_cornell_config0_cpu_plumbing():
    """BASELINE.json configs[0] on the CPU only (no GPU): Cornell box 256x256, 16 spp through the oracle — finite,
    lit, red wall on the left / green on the right, sky visible through the open front."""
    from conftest import CORNELL_CAM
    sc, buf = oracle_scene("cornell.xml")
    assert sc.prim_count == 12 and sc.triangle_count == 10
    u = ob.make_uniforms(256, 256, sc.prim_count, sc.triangle_count, cam=CORNELL_CAM)
    img, ct = ob.render(u, buf, rng_mode=ob.RNG_PHILOX, max_depth=32, accumulate=1, sample_count=16, seed=(1, 0), threads=8)
    img = img / 16
    assert np.isfinite(img).all() and ct["paths"] == 256 * 256 * 16
    left, right = img[100:180, 20:50, :3].mean((0, 1)), img[100:180, 206:236, :3].mean((0, 1))
    assert left[0] > 2 * left[1] and right[1] > 2 * right[0]      # red wall / green wall
    assert ct["emissive_hits"] > 0 and ct["misses"] > 0


def test_polygon_soup_ingest(tmp_path):
    """OBJ polygons: product loader and oracle against what the reference's tinyobjloader 2.0.0 returns for the
    committed input (fixture made by tests/golden/make_ingest_golden.py with the real library)."""
    import ctypes as C
    from metalpathtracer_amd import host
    exp = np.load(os.path.join(GOLDEN, "polygon_soup_expected.npz"))
    verts, tris = exp["verts"], exp["tris"]
    want = np.stack([verts[tris[:, k]] for k in range(3)], 1)   # position 0, scale 1: 0 + 1*v == v bit for bit
    want = np.float32(0) + np.float32(1) * want
    xml = tmp_path / "soup.xml"
    xml.write_text('<Scene><Mesh file="%s" position="0,0,0" scale="1" albedo="1,1,1" emission="0,0,0"/></Scene>'
                   % os.path.join(GOLDEN, "polygon_soup.obj"))
    hs = host.Scene()
    st, _ = host.SceneLoader.LoadSceneFromXML(str(xml), hs)
    assert st == 0
    osn = ob.OracleScene()
    assert osn.load_xml(str(xml)) == 0
    op = np.zeros((osn.prim_count, 3, 4), np.float32)
    ob.lib().orc_scene_pack_prims(osn.h, op.ctypes.data_as(C.POINTER(C.c_float)))
    for got in (hs.buffers()[1], op):
        assert got.shape[0] == len(tris)
        np.testing.assert_array_equal(got[:, :, :3].view(np.uint32), want.view(np.uint32))
