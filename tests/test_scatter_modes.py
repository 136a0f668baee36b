"""The three BSDF modes of the bounce (include/mpt.h): rayColor's own Lambert bounce (PathTracing.h:251-255), Scatter.h's
mirror / dielectric branches (Scatter.h:28-40) and Scatter.h's own Lambert branch (Scatter.h:24-27 with randomFloat3 of
Random.h:18-30), which is dead code in the reference.  CPU only: the oracle against a numpy restatement written from
the reference text, on a scene whose answer has a closed form."""
import numpy as np
import pytest

from oracle import binding as ob

f32 = np.float32


def pcg_hash(s):                      # Random.h:6-11
    state = (int(s) * 747796405 + 2891336453) & 0xFFFFFFFF
    word = ((state >> ((state >> 28) + 4)) ^ state) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


def pcg_float(s):                     # Random.h:13-16
    return f32(pcg_hash(s)) / f32(4294967295)


def normalize(v):
    v = np.asarray(v, f32)
    return v / f32(np.sqrt(f32(v[0] * v[0]) + f32(v[1] * v[1]) + f32(v[2] * v[2])))


@pytest.fixture(scope="module")
def ground(tmp_path_factory):
    """One diffuse square at y = 0 seen from straight above: every primary ray hits it, every bounce leaves to the sky."""
    d = tmp_path_factory.mktemp("ground")
    # (a third triangle out of sight gives the leaf's box a thickness: the reference's slab test never enters a flat box)
    (d / "quad.obj").write_text("v -1000 0 -1000\nv 1000 0 -1000\nv 1000 0 1000\nv -1000 0 1000\nv 2000 -5 2000\n"
                                "v 2001 -5 2000\nv 2000 -5 2001\nf 1 2 3\nf 1 3 4\nf 5 6 7\n")
    (d / "ground.xml").write_text('<Scene>\n  <Mesh file="quad.obj" position="0,0,0" scale="1" albedo="0.5,0.25,0.75" '
                                  'emission="0,0,0" materialType="0" emissionPower="0"/>\n</Scene>\n')
    sc = ob.OracleScene()
    assert sc.load_xml(str(d / "ground.xml")) == 0
    sc.build_bvh()
    cam = dict(pos=(0.0, 10.0, 0.0), fwd=(0.0, -1.0, 0.0), up=(0.0, 0.0, -1.0), vfov=40.0)
    return sc, sc.buffers(), cam


def test_scatter_all_diffuse_bounce_follows_random_float3(ground):
    """Literal RNG with randomSeed = 0: every pixel enters rayColor with the same stuck seed, so the bounce
    off the ground is ONE direction, n + normalize(cube point), and the pixel is albedo * sky(direction) — half of it
    after the first frame (frameCount starts at 1)."""
    sc, buf, cam = ground
    W, H = 24, 16
    u = ob.make_uniforms(W, H, sc.prim_count, sc.triangle_count, cam=cam, random_seed=(0.0, 0.0, 0.0), frame_count=1)
    s = pcg_hash(pcg_hash(0))         # Fragment.metal:31-34: two draws for the jitter, then rayColor's seed
    c = []
    for _ in range(3):                # Random.h:18-30
        c.append(pcg_float(s) * f32(2) - f32(1))
        s = pcg_hash(s)
    n = np.array([0, 1, 0], f32)
    want = {}
    d = normalize(n + normalize(c))   # Scatter.h:26,42
    want[ob.BSDF_SCATTER_ALL] = d
    z = f32(2) * pcg_float(pcg_hash(pcg_hash(0))) - f32(1)     # PathTracing.h:25-31: the same u for z and phi
    t = f32(2) * f32(3.14159274101257324) * pcg_float(pcg_hash(pcg_hash(0)))
    rr = np.sqrt(f32(1) - z * z)
    want[ob.BSDF_LAMBERT] = normalize(n + np.array([rr * np.cos(t), rr * np.sin(t), z], f32))
    want[ob.BSDF_SCATTER] = want[ob.BSDF_LAMBERT]              # a diffuse surface keeps rayColor's bounce in that mode
    assert abs(float(want[ob.BSDF_SCATTER_ALL][1]) - float(want[ob.BSDF_LAMBERT][1])) > 1e-3
    albedo = np.array([0.5, 0.25, 0.75], f32)
    for mode, d in want.items():
        img, ct = ob.render(u, buf, rng_mode=ob.RNG_LITERAL, bsdf_mode=mode, max_depth=32, accumulate=0)
        assert ct["rays"] == 2 * W * H and ct["misses"] == W * H
        tt = f32(0.5) * (d[1] + f32(1))
        sky = np.array([1, 1, 1], f32) + (np.array([0.6, 0.7, 1.0], f32) - f32(1)) * tt          # PathTracing.h:227-231
        np.testing.assert_allclose(img[..., :3], np.broadcast_to(f32(0.5) * albedo * sky, (H, W, 3)), rtol=2e-6, atol=0)


def test_modes_agree_until_a_bounce_direction_is_used(ground):
    sc, buf, cam = ground
    u = ob.make_uniforms(32, 20, sc.prim_count, sc.triangle_count, cam=cam)
    kw = dict(rng_mode=ob.RNG_PHILOX, accumulate=1, sample_count=4, seed=(5, 9))
    one = [ob.render(u, buf, bsdf_mode=m, max_depth=1, **kw)[0] for m in (0, 1, 2)]
    np.testing.assert_array_equal(one[0], one[1])
    np.testing.assert_array_equal(one[0], one[2])
    two = [ob.render(u, buf, bsdf_mode=m, max_depth=2, **kw)[0] for m in (0, 1, 2)]
    np.testing.assert_array_equal(two[0], two[1])              # no mirror or glass in this scene
    assert not np.array_equal(two[0], two[2])
    # the cube-point direction is not cosine-distributed, but it is still a hemisphere bounce off an open ground: all sky
    assert np.isfinite(two[2]).all() and two[2][..., :3].min() > 0
