"""GPU tests that close round 2's parity holes.

1. configs[3] at its real size: bunny x20 at 3840x2160, one GPU's 1/8 tile shard (rank 7), 1030 spp so that the render
   crosses a pass boundary by itself (1024 + 6 samples: 1.06 G paths per pass, 17 GB of result slots) — determinism,
   closest-first == reference order, additivity in samples across that boundary; at low spp additivity over all 8 shards
   and a 32-row band against the oracle.
2. The closest-first rule under attack.  mpt_ordered.h culls sub-trees that start beyond best t * (1 + 2^-10) + eps; that
   is exact unless a culled triangle's COMPUTED t lies in front of its own box by more than the margin, which needs a ray
   within ~1e-5 / |e1 x e2| of the triangle's plane (|det| just above the reference's 1e-5 threshold, PathTracing.h:153).
   The scenes below are built to go there: slivers and needles hit at grazing angles from far away, triangles 10^3..10^5
   units across next to millimetre ones, origins at and beyond o_limit, 16 and 17 spheres, spheres inside meshes,
   coplanar duplicates.  mpt_trace_rays_ordered must return exactly what mpt_trace_rays (the reference-order walk) returns
   for >= 1e8 rays — except on rays where the reference's answer is itself an arithmetic artefact, a hit in front of the
   hit triangle's own bounding box, which the test identifies in exact arithmetic and counts (< 1e-4 of the rays of any
   family; none on well-conditioned meshes) — and the re-trace flags must stay in their bands (a flag rate that explodes would hide a slow path;
   one that vanishes would mean the scene does not reach the rule it is aimed at)."""
import numpy as np
import pytest

from oracle import binding as ob
from test_gpu_parity import setup, setup_tree

pytestmark = pytest.mark.gpu


def _same(a, b):
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("tree", ["reference", "device"])
def test_config3_4k_shard_crosses_a_pass_boundary(gpu_ctx, tree):
    from metalpathtracer_amd import capi
    W, H = 3840, 2160
    buf, uo = setup_tree(gpu_ctx, "bunny20.xml", W, H, tree)   # (both tree routes: the device-built tree is what --bvh auto renders)
    kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, seed=(1, 0))
    shard = dict(shard_rank=7, shard_count=8)
    # ---- the config's own size: 1/8 of the tiles, 1030 spp = passes of 1024 + 6 samples --------------------------------
    gpu_ctx.clear_sum()
    gpu_ctx.reset_stats()
    gpu_ctx.render(sample_count=1030, **shard, **kw)
    st = gpu_ctx.stats()
    assert st["trace_launches"] == 2                      # the pass limit (2^30 paths) split the render, nothing else did
    assert st["paths"] == 16200 * 64 * 1030               # 480 x 270 tiles / 8 ranks, every pixel inside the image
    assert st["tree_parked"] > 0                          # AUTO = the closest-first pipeline at this scene size
    a = gpu_ctx.read_sum()
    assert np.isfinite(a).all() and a[..., 3].max() > 0
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_count=1030, **shard, **kw)
    _same(a, gpu_ctx.read_sum())                                                  # deterministic
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_begin=0, sample_count=1024, **shard, **kw)
    gpu_ctx.render(sample_begin=1024, sample_count=6, **shard, **kw)
    _same(a, gpu_ctx.read_sum())                                                  # additive across the pass boundary
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_count=1030, pipeline=capi.PIPE_WAVELOCAL, **shard, **kw)
    _same(a, gpu_ctx.read_sum())                                                  # closest-first == reference order
    owned = a[..., 3] > 0                                 # alpha counts sky hits and emitters: every owned pixel has some
    ty, tx = np.nonzero(owned)
    assert set(((ty // 8) * 480 + tx // 8) % 8) == {7}    # only rank 7's tiles were touched
    del a
    # ---- low spp: the 8 shards add up to the unsharded image; a band of rows against the oracle ---------------------------
    gpu_ctx.clear_sum()
    gpu_ctx.render(sample_count=2, **kw)
    whole = gpu_ctx.read_sum()
    total = np.zeros_like(whole)
    for r in range(8):
        gpu_ctx.clear_sum()
        gpu_ctx.render(sample_count=2, shard_rank=r, shard_count=8, **kw)
        total += gpu_ctx.read_sum()
    _same(whole, total)
    rows = (1120, 1152)                                   # through the bunnies
    ref, _ = ob.render(uo, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=2, seed=(1, 0), rows=rows, threads=8)
    _same(whole[rows[0]:rows[1]], ref[rows[0]:rows[1]])


# ---- adversarial scenes ----------------------------------------------------------------------------------------------------
def _rand_unit(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _sliver_scene(rng, n_tri, size_lo, size_hi, aspect_lo, aspect_hi, spread):
    """n_tri thin triangles: long edge log-uniform in [size_lo, size_hi], width = long / aspect, random orientation,
    centres normal(0, spread)."""
    c = rng.normal(size=(n_tri, 3)) * spread
    long_ = np.exp(rng.uniform(np.log(size_lo), np.log(size_hi), n_tri))
    aspect = np.exp(rng.uniform(np.log(aspect_lo), np.log(aspect_hi), n_tri))
    a = _rand_unit(rng, n_tri)
    b = np.cross(a, _rand_unit(rng, n_tri))
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    v0 = c - 0.5 * long_[:, None] * a
    v1 = c + 0.5 * long_[:, None] * a
    v2 = c + (long_ / aspect)[:, None] * b + rng.uniform(-0.5, 0.5, (n_tri, 1)) * long_[:, None] * a
    return np.stack([v0, v1, v2], 1).astype(np.float32)


def _grid_scene(rng, n, cell, jitter):
    """A height field of n x n cells (2 n^2 triangles) with millimetre-to-unit jitter: neighbours share edges exactly."""
    xs = (np.arange(n + 1) - n / 2) * cell
    h = rng.uniform(-jitter, jitter, (n + 1, n + 1))
    P = np.stack(np.broadcast_arrays(xs[None, :], h, xs[:, None]), -1).astype(np.float32)
    t = []
    for i in range(n):
        for j in range(n):
            t.append((P[i, j], P[i, j + 1], P[i + 1, j]))
            t.append((P[i, j + 1], P[i + 1, j + 1], P[i + 1, j]))
    return np.asarray(t, np.float32)


def _build(tris, spheres=(), mode=0):
    from metalpathtracer_amd import host
    sc = host.Scene()
    for c, r in spheres:
        sc.addSphere([float(x) for x in c], float(r))
    for t in tris:
        sc.addTriangle([float(x) for x in t[0]], [float(x) for x in t[1]], [float(x) for x in t[2]])
    sc.buildBVH(mode)
    return sc, sc.buffers()


def _grazing_rays(rng, tris, n, theta_lo, theta_hi, dist_lo, dist_hi):
    """Rays aimed at random points of random triangles, nearly IN the triangle's plane: the angle between ray and plane is
    log-uniform in [theta_lo, theta_hi] (either side), the origin dist_lo..dist_hi away (log-uniform) — so |det| =
    |d . (e2 x e1)| lands on both sides of the reference's 1e-5 threshold, and other triangles lie in front and behind."""
    k = rng.integers(0, len(tris), n)
    t = tris[k].astype(np.float64)
    e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
    u = rng.uniform(0, 1, n)
    v = rng.uniform(0, 1, n)
    flip = u + v > 1
    u[flip], v[flip] = 1 - u[flip], 1 - v[flip]
    x = t[:, 0] + u[:, None] * e1 + v[:, None] * e2
    nrm = np.cross(e1, e2)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    inpl = np.cross(nrm, _rand_unit(rng, n))
    inpl /= np.maximum(np.linalg.norm(inpl, axis=1, keepdims=True), 1e-300)
    theta = np.exp(rng.uniform(np.log(theta_lo), np.log(theta_hi), n)) * rng.choice([-1.0, 1.0], n)
    d = inpl * np.cos(theta)[:, None] + nrm * np.sin(theta)[:, None]
    dist = np.exp(rng.uniform(np.log(dist_lo), np.log(dist_hi), n))
    o = x - d * dist[:, None]
    return o.astype(np.float32), d.astype(np.float32)


def _random_rays(rng, n, spread, inside_frac=0.3):
    o = (rng.normal(size=(n, 3)) * spread * 2.0).astype(np.float32)
    tgt = rng.normal(size=(n, 3)) * spread
    d = (tgt - o)
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    k = int(n * inside_frac)
    o[:k] *= 0.2
    return o, d.astype(np.float32)


def _artefact(prims, pid, o, d, t):
    """True if the hit (primitive pid at computed distance t) is geometrically impossible: in front of the primitive's own
    bounding box by more than the culling margin t * 2^-10 (or the ray misses that box altogether).  Exact arithmetic."""
    p = prims[pid].astype(np.float64)
    if p[0, 3] != 1.0:
        return False
    v = np.stack([p[0, :3], p[1, :3], p[2, :3]])
    lo, hi = v.min(0), v.max(0)
    o, d = o.astype(np.float64), d.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = (lo - o) / d, (hi - o) / d
    t_in, t_out = np.nanmax(np.minimum(t0, t1)), np.nanmin(np.maximum(t0, t1))
    return bool(t_in > t_out or float(t) < t_in - (abs(t_in) * 2.0 ** -10 + 1e-6 * np.abs(v).max()))


def _compare(ctx, o, d, hist, prims):
    """mpt_trace_rays_ordered against mpt_trace_rays.  The closest-first rule is exact except where the REFERENCE's answer is
    itself an artefact of its arithmetic — a triangle accepted at a computed t in front of the triangle's own bounding box
    (include/mpt.h, MPT_PIPE_ORDERED): every difference must be of that kind, and they must be rare."""
    t0, p0, n0, f0 = ctx.trace_rays(o, d)
    t1, p1, n1, f1, fl = ctx.trace_rays_ordered(o, d)
    bad = np.nonzero((t0.view(np.uint32) != t1.view(np.uint32)) | (p0 != p1))[0]
    for i in bad:
        assert p0[i] >= 0 and _artefact(prims, p0[i], o[i], d[i], t0[i]), (
            "closest-first != reference order on a ray whose reference answer is no artefact: ray %d o=%r d=%r  ref (t=%r prim=%d)  got (t=%r prim=%d flags=%d)"
            % (i, o[i].tolist(), d[i].tolist(), float(t0[i]), int(p0[i]), float(t1[i]), int(p1[i]), int(fl[i])))
    ok = np.ones(len(o), bool)
    ok[bad] = False
    _same(n0[ok], n1[ok])
    np.testing.assert_array_equal(f0[ok], f1[ok])
    for bit in (1, 2, 4, 8):
        hist[bit] = hist.get(bit, 0) + int(((fl & bit) != 0).sum())
    hist["n"] = hist.get("n", 0) + len(o)
    hist["hits"] = hist.get("hits", 0) + int((p0 >= 0).sum())
    hist["artefacts"] = hist.get("artefacts", 0) + int(bad.size)


def _dump_report(mode, report):
    """flag histograms per scene family, next to the other GPU-run artefacts (for tuning the bands below)"""
    import json, os
    from conftest import ROOT
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        json.dump({k: {kk: ({str(b): c for b, c in vv.items()} if isinstance(vv, dict) else vv) for kk, vv in v.items()}
                   for k, v in report.items()}, open(os.path.join(d, "adversarial_flags_mode%d.json" % mode), "w"), indent=1)


CASES = [
    # name, triangles, spheres, ray recipe (grazing angle range, distance range), random-ray spread
    ("slivers 1..30 units, aspect 10..1e4", lambda r: _sliver_scene(r, 12000, 1.0, 30.0, 10.0, 1e4, 25.0), (), (1e-8, 1e-2, 1.0, 300.0), 40.0),
    ("needles 0.01..2 units, aspect 1e2..1e5", lambda r: _sliver_scene(r, 12000, 0.01, 2.0, 1e2, 1e5, 3.0), (), (1e-7, 1e-1, 0.1, 60.0), 5.0),
    ("huge 1e3..1e5 next to millimetres", lambda r: np.concatenate([_sliver_scene(r, 40, 1e3, 1e5, 1.0, 30.0, 50.0),
                                                                   _sliver_scene(r, 12000, 1e-3, 1e-1, 1.0, 8.0, 2.0)]), (), (1e-9, 1e-2, 0.5, 2000.0), 4.0),
    ("height field, millimetre jitter", lambda r: _grid_scene(r, 80, 0.5, 1e-3), (), (1e-7, 1e-2, 0.5, 80.0), 25.0),
    ("height field, unit jitter, spheres inside", lambda r: _grid_scene(r, 80, 0.5, 0.7),
     tuple(((x, 0.0, z), 0.9) for x in (-12.0, -4.0, 4.0, 12.0) for z in (-12.0, -4.0, 4.0, 12.0)), (1e-6, 1e-1, 0.5, 80.0), 25.0),
    ("coplanar duplicates", lambda r: np.repeat(_sliver_scene(r, 4000, 0.5, 6.0, 1.0, 20.0, 8.0), 3, axis=0), (), (1e-6, 1.0, 0.5, 60.0), 12.0),
]


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_closest_first_rule_under_attack(gpu_ctx, mode):
    """Three tree builders x six scene families x (grazing + random + far-origin rays): ~1.1e8 rays in all, bit-identical
    between the closest-first walk and the reference-order walk except for the reference's own artefacts (_compare)."""
    from metalpathtracer_amd import capi
    rng = np.random.default_rng(100 + mode)
    total = 0
    B = 1 << 21
    report = {}
    for name, make, spheres, (th_lo, th_hi, d_lo, d_hi), spread in CASES:
        tris = make(rng)
        sc, buf = _build(tris, spheres, mode)
        gpu_ctx.upload_scene(*buf)
        info = gpu_ctx.accel_info()
        assert info["ordered_ok"] == 1 and info["always_spheres"] == len(spheres), name
        hist = {}
        for rep in range(2):
            o, d = _grazing_rays(rng, tris, B, th_lo, th_hi, d_lo, d_hi)
            _compare(gpu_ctx, o, d, hist, buf[1])
        o, d = _random_rays(rng, B, spread)
        _compare(gpu_ctx, o, d, hist, buf[1])
        # origins at o_limit (64 x the largest |coordinate| of a triangle vertex) and beyond: exactly there, 1 ulp either side, 10x
        ext = float(np.abs(tris).max())
        o, d = _random_rays(rng, B // 8, spread, inside_frac=0.0)
        scale = np.asarray([64.0, np.nextafter(np.float32(64.0), np.float32(0)), np.nextafter(np.float32(64.0), np.float32(100)), 640.0], np.float32)
        axis = rng.integers(0, 3, len(o))
        o[np.arange(len(o)), axis] = ext * scale[rng.integers(0, 4, len(o))] * rng.choice([-1.0, 1.0], len(o)).astype(np.float32)
        tgt = rng.normal(size=o.shape) * spread
        d = tgt - o
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        far = {}
        _compare(gpu_ctx, o, d, far, buf[1])
        assert far[1] >= far["n"] // 4, (name, far)           # beyond the limit: handed to the reference-order walk
        total += hist["n"] + far["n"]
        n = hist["n"]
        report[name] = {"rays": hist, "far_origin_rays": far, "own_nodes": info["nodes"], "depth": info["depth"]}
        _dump_report(mode, report)
        # flags: the rule's escape hatches are used, none of them floods
        assert hist["artefacts"] <= 1e-4 * n, (name, hist)    # reference answers in front of their own triangle's box: rare even here
        assert hist[1] <= 0.02 * n, (name, hist)              # (nearly) axis-parallel directions are rare in these rays
        assert hist[4] <= 0.25 * n, (name, hist)              # winners in front of their own leaf box: grazing hits do that
        # (flag 8, stack overflow with 8 entries, is the rule rather than the exception for piles of overlapping slivers and for
        #  millimetre triangles inside 1e5-unit ones: recorded in the report, not bounded)
        assert hist["hits"] >= 0.02 * n, (name, hist)
        if name == "coplanar duplicates":
            assert hist[2] >= 0.2 * hist["hits"], (name, hist)  # exact ties between the copies
    assert total >= 3.6e7


def test_sixteen_spheres_qualify_seventeen_do_not(gpu_ctx):
    """The always list holds at most 16 spheres: with 17 the scene falls back to the reference-order pipeline."""
    from metalpathtracer_amd import capi
    rng = np.random.default_rng(7)
    tris = _grid_scene(rng, 40, 0.5, 0.3)
    for n_sph, ok in ((16, 1), (17, 0)):
        spheres = tuple(((float(x), 1.0, float(z)), 0.8) for x, z in rng.uniform(-9, 9, (n_sph, 2)))
        sc, buf = _build(tris, spheres)
        gpu_ctx.upload_scene(*buf)
        info = gpu_ctx.accel_info()
        assert info["ordered_ok"] == ok and info["auto_pipeline"] == capi.PIPE_WAVELOCAL     # (small scene either way)
        if ok:
            hist = {}
            o, d = _random_rays(rng, 1 << 20, 10.0)
            _compare(gpu_ctx, o, d, hist, buf[1])
            assert info["always_spheres"] == 16 and hist["hits"] > hist["n"] // 4


def test_auto_keeps_the_reference_order_for_the_literal_rng(gpu_ctx):
    """MPT_PIPE_AUTO: the closest-first pipeline for a big scene with the philox RNG (the product's own batch mode), the
    reference-order pipeline with the literal RNG (the mode that exists to reproduce the reference's frames)."""
    from metalpathtracer_amd import capi
    setup(gpu_ctx, "bunny20.xml", 160, 90)
    assert gpu_ctx.accel_info()["auto_pipeline"] == capi.PIPE_ORDERED
    gpu_ctx.reset_stats()
    gpu_ctx.draw(rng_mode=capi.RNG_LITERAL, max_depth=8)
    assert gpu_ctx.stats()["tree_parked"] == 0 and gpu_ctx.stats()["rays"] > 0
    gpu_ctx.clear_sum()
    gpu_ctx.reset_stats()
    gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=1)
    assert gpu_ctx.stats()["tree_parked"] > 0


@pytest.mark.parametrize("case", range(len(CASES)))
def test_adversarial_scenes_render_the_same_image_on_both_walks(gpu_ctx, case):
    """north_star's gate on the one class of rays where the two walks may differ (the reference's own grazing artefacts,
    include/mpt.h MPT_PIPE_ORDERED): every adversarial family rendered as an IMAGE — camera inside the cloud, 640x360 x 16 spp,
    depth 8, sky light — once by the closest-first pipeline (what MPT_PIPE_AUTO runs for a scene of this size with the philox
    RNG: asserted) and once by the reference-order pipeline (k_wavelocal).  Stated tolerance: per-pixel L2 < 1e-3 on the resolved
    image; the number of pixels that differ at all, the largest difference and the ray counts are printed (and asserted small:
    a differing ray changes one of a pixel's 16 samples).  The device-built tree (what --bvh auto renders) and the reference's
    own tree are both rendered."""
    from metalpathtracer_amd import capi, host
    name, make, spheres, _, spread = CASES[case]
    rng = np.random.default_rng(500 + case)
    tris = make(rng)
    W, H, spp = 640, 360, 16
    for tree in (host.BVH_DEVICE, host.BVH_REFERENCE_SWEEP):
        sc = host.Scene()
        for c, r in spheres:
            sc.addSphere([float(x) for x in c], float(r), albedo=(0.9, 0.6, 0.3))
        for k, t in enumerate(tris):       # every fifth triangle emits (scenes that enclose the camera still make a picture), the rest are diffuse
            kw = dict(albedo=(0.2, 0.2, 0.2), emission=(1.0, 0.8, 0.6), emissionPower=1.5) if k % 5 == 0 else \
                 dict(albedo=(0.8, 0.8, 0.8) if k % 7 else (0.9, 0.3, 0.2))
            sc.addTriangle([float(x) for x in t[0]], [float(x) for x in t[1]], [float(x) for x in t[2]], **kw)
        host.make_ready(gpu_ctx, sc, tree)
        info = gpu_ctx.accel_info()
        assert info["ordered_ok"] == 1 and info["auto_pipeline"] == capi.PIPE_ORDERED, (name, info)
        cam = dict(pos=(0.0, 0.05 * spread, 0.0), fwd=(0.28603878, -0.09534626, -0.95346259), up=(0.0, 1.0, 0.0), vfov=70.0)
        gpu_ctx.resize(W, H)
        gpu_ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), cam=cam))
        img, rays = {}, {}
        for pipe in (capi.PIPE_AUTO, capi.PIPE_WAVELOCAL):
            gpu_ctx.clear_sum()
            gpu_ctx.reset_stats()
            gpu_ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=spp, seed=(3, 9), pipeline=pipe)
            img[pipe] = gpu_ctx.read_sum() / spp
            st = gpu_ctx.stats()
            rays[pipe] = st["rays"]
            assert (st["tree_parked"] > 0) == (pipe == capi.PIPE_AUTO), (name, pipe, st)   # AUTO really ran k_ordered
        a, b = img[capi.PIPE_AUTO], img[capi.PIPE_WAVELOCAL]
        assert np.isfinite(a).all() and np.isfinite(b).all()
        d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
        l2 = float(np.sqrt((d * d).sum(-1).mean()))
        npix = int((np.abs(d).max(-1) > 0).sum())
        bounce = rays[capi.PIPE_WAVELOCAL] - W * H * spp      # rays beyond the primaries = surface hits that went on
        print("%-45s tree %d: per-pixel L2 %.3g, pixels that differ %d of %d, max |d| %.3g, rays %d vs %d (%d bounce rays), image std %.3g"
              % (name, tree, l2, npix, W * H, float(np.abs(d).max()), rays[capi.PIPE_AUTO], rays[capi.PIPE_WAVELOCAL], bounce, float(b[..., :3].std())))
        assert l2 < 1e-3, (name, tree, l2, npix)
        assert npix <= 1e-4 * W * H * spp, (name, tree, npix)          # (observed: 0 — natural rays do not graze at 1e-5 rad)
        assert abs(rays[capi.PIPE_AUTO] - rays[capi.PIPE_WAVELOCAL]) <= 1e-5 * rays[capi.PIPE_WAVELOCAL] + 8
        assert bounce >= 40000 and b[..., :3].std() > 1e-3, name       # surfaces were hit (needles: 1 % of the paths; an enclosing scene: all) and it is a picture
