"""Pins the CPU oracle to the values SURVEY.md App. C records from the reference's own shader / BVH text.

The reference has no tests or golden vectors of its own; these recorded values are the only
reference-derived numbers available.  Integer known answers are exact; pixel values are the 6 digits the
survey printed; per-ray work counters agree to <= 5e-5 relative (MSL normalize() is implementation-defined
to the ulp, see oracle/mpt_oracle.cpp) with the ulp-independent ones (emissive hits, sky misses) exact.
"""
import numpy as np
import pytest

from conftest import oracle_scene
from oracle import binding as ob


def test_pcg_known_answers():  # SURVEY App. C.1  <- R/Renderer/Shaders/Random.h:6-16
    L = ob.lib()
    assert [L.orc_pcg_hash(x) for x in (0, 1, 2, 12345, 0xFFFFFFFF)] == [
        2891249901, 3639127469, 86804957, 261270601, 2144086741]
    assert L.orc_pcg_float(0) == pytest.approx(0.67317158, abs=5e-9)
    assert L.orc_pcg_float(1) == pytest.approx(0.84730041, abs=5e-9)
    assert L.orc_pcg_float(2) == pytest.approx(0.0202108547, abs=5e-10)


def test_seed_chain_from_zero():  # u0, seed, u1, seed, stuck vector
    L = ob.lib()
    assert L.orc_pcg_float(0) == pytest.approx(0.6731716, abs=5e-8)
    s1 = L.orc_pcg_hash(0)
    assert s1 == 2891249901
    assert L.orc_pcg_float(s1) == pytest.approx(0.7159897, abs=5e-8)
    s2 = L.orc_pcg_hash(s1)
    assert s2 == 3075152351
    u = L.orc_pcg_float(s2)
    assert u == pytest.approx(0.2613313, abs=5e-8)
    z = np.float32(2.0) * np.float32(u) - np.float32(1.0)
    t = np.float32(2.0) * np.float32(3.14159274) * np.float32(u)
    r = np.sqrt(np.float32(1.0) - z * z)
    v = (r * np.cos(t), r * np.sin(t), z)
    assert v == pytest.approx((-0.0625090, 0.8764939, -0.4773374), abs=2e-7)


def test_host_seed_stream():  # SURVEY App. C.1  <- R/Renderer/Renderer.cpp:30-41
    import ctypes as C
    st = C.c_uint32(92407235)
    got = [ob.lib().orc_bitm_random(C.byref(st)) for _ in range(4)]
    assert got == [3475558128, 1633319866, 404750907, 100482832]
    assert ob.host_seed_sequence(3) == pytest.approx([0.80921644, 0.38028690, 0.09423842], abs=5e-9)


def test_uniforms_layout():  # 144 bytes, offsets
    import ctypes as C
    U = ob.Uniforms
    assert C.sizeof(U) == 144
    off = {n: getattr(U, n).offset for n, _ in U._fields_}
    assert (off["cameraPosition"], off["screenSize"], off["viewportU"], off["viewportV"], off["firstPixelPosition"],
            off["randomSeed"], off["primitiveCount"], off["triangleCount"], off["frameCount"],
            off["totalPrimitiveCount"]) == (16, 32, 48, 64, 80, 96, 112, 120, 128, 136)


def test_camera_viewport_values():
    u = ob.make_uniforms(1280, 720, 1)
    assert list(u.viewportU)[:3] == pytest.approx([2.0528, 0, 0], abs=5e-5)
    assert list(u.viewportV)[:3] == pytest.approx([0, -1.1547, 0], abs=5e-5)
    assert list(u.firstPixelPosition)[:3] == pytest.approx([-1.0264, 20.5774, 49.0], abs=5e-5)


def test_scene_xml_bvh_shape():
    sc, (bvh, prims, mats, idx) = oracle_scene("scene.xml")
    assert sc.prim_count == 4971 and sc.triangle_count == 4968 and sc.node_count == 1789
    count = bvh[:, 1, 3].view(np.int32)
    assert (count > 0).sum() == 895 and count.max() == 8
    assert prims[:3, 0, 3].tolist() == [0, 0, 0] and (prims[3:, 0, 3] == 1).all()  # spheres first
    np.testing.assert_array_equal(bvh[0, 0, :3], [-10000, -20000, -10000])
    np.testing.assert_array_equal(bvh[0, 1, :3], [10000, 140, 10000])
    assert bvh[0, 0, 3].view(np.int32) == 1 and count[0] == -2


def test_oracle_image_seed0_640x360():  # SURVEY App. C.3 (frame 1 => 1/2 c)
    sc, buf = oracle_scene("scene.xml")
    u = ob.make_uniforms(640, 360, sc.prim_count, sc.triangle_count, frame_count=1)
    img, _ = ob.render(u, buf, rng_mode=ob.RNG_LITERAL, accumulate=0, threads=8)
    mean = img[..., :3].astype(np.float64).reshape(-1, 3).mean(0)
    assert mean == pytest.approx([0.336689, 0.360850, 0.441971], abs=1e-6)
    assert img[0, 0, :3] == pytest.approx([0.362723, 0.397042, 0.5], abs=1e-6)
    assert img[180, 160, :3] == pytest.approx([0.400204, 0.425153, 0.5], abs=1e-6)
    assert img[180, 320, :3].tolist() == [0.5, 0.5, 0.5]  # light sphere centre


def test_oracle_image_host_seed_1080p():
    sc, buf = oracle_scene("scene.xml")
    u = ob.make_uniforms(1920, 1080, sc.prim_count, sc.triangle_count, random_seed=ob.host_seed_sequence(3),
                         frame_count=1)
    img, ct = ob.render(u, buf, rng_mode=ob.RNG_LITERAL, accumulate=0, threads=8)
    mean = img[..., :3].astype(np.float64).reshape(-1, 3).mean(0)
    assert mean == pytest.approx([0.341791, 0.365638, 0.446549], abs=1e-6)
    assert img[0, 0, :3] == pytest.approx([0.362659, 0.396995, 0.5], abs=1e-6)
    assert img[720, 640, :3] == pytest.approx([0.261582, 0.296186, 0.4], abs=1e-6)
    assert ct["rays"] / ct["paths"] == pytest.approx(1.687, abs=1e-3)  # 1.687 rays/path


def test_oracle_work_counters_720p():  # SURVEY App. C.3 gcov counts, host seed, 1280x720
    sc, buf = oracle_scene("scene.xml")
    u = ob.make_uniforms(1280, 720, sc.prim_count, sc.triangle_count, random_seed=ob.host_seed_sequence(3),
                         frame_count=1)
    _, ct = ob.render(u, buf, rng_mode=ob.RNG_LITERAL, accumulate=0, threads=8)
    assert ct["paths"] == 921600
    assert ct["emissive_hits"] == 74701          # exact
    assert ct["misses"] == 920323                # exact
    want = dict(rays=1555067, node_pops=13145685, aabb_pass=8490497, prim_tests=6152603, sphere_tests=3560094,
                tri_tests=2592509, bounces=634744)
    for k, v in want.items():
        assert abs(ct[k] - v) / v < 5e-5, (k, ct[k], v)
    assert abs(ct["pushes"] // 2 - 5795309) / 5795309 < 5e-5  # the survey counted line 191 (one per internal hit)


def test_running_mean_four_frames():  # 4 frames @320x180 -> light pixel 0.8 (= 4/5)
    sc, buf = oracle_scene("scene.xml")
    W, H = 320, 180
    last = np.zeros((H, W, 4), np.float32)
    vals = []
    for f in range(1, 5):
        u = ob.make_uniforms(W, H, sc.prim_count, sc.triangle_count, frame_count=f)
        cur, _ = ob.render(u, buf, rng_mode=ob.RNG_LITERAL, accumulate=0, last=last, threads=8)
        vals.append(float(cur[90, 160, 0]))
        last = cur
    assert vals[0] == 0.5 and vals[-1] == pytest.approx(0.8, abs=1e-6)


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32_10
    import ctypes as C
    cases = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
             ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
             ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
              (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in cases:
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        ob.lib().orc_philox(c, k, o)
        assert tuple(o) == want


def test_sincos_accuracy():
    import ctypes as C
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for u in np.linspace(0.0, 1.0, 4097, dtype=np.float32)[:-1]:
        ob.lib().orc_sincos_2pi(float(u), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(2 * np.pi * float(u))), abs(c.value - np.cos(2 * np.pi * float(u))))
    assert worst < 2e-7


def test_philox_workload_matches_survey_proxy():  # advancing-RNG proxy, scene.xml
    sc, buf = oracle_scene("scene.xml")
    u = ob.make_uniforms(640, 360, sc.prim_count, sc.triangle_count)
    _, ct = ob.render(u, buf, rng_mode=ob.RNG_PHILOX, accumulate=1, sample_count=1, threads=8)
    assert ct["rays"] / ct["paths"] == pytest.approx(1.682, abs=0.01)
    assert ct["node_pops"] / ct["rays"] == pytest.approx(7.52, abs=0.05)
    assert ct["prim_tests"] / ct["rays"] == pytest.approx(3.59, abs=0.03)
    assert ct["bounces"] / ct["rays"] == pytest.approx(0.405, abs=0.005)
