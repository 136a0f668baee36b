"""GPU tests of the BVH built ON the GPU (mpt_build_bvh / Scene::buildBVH(GpuLbvh)): the arrays are a
well-formed tree in the reference's format with exactly nested boxes and leaves of <= 8 primitives, and — the tree is not
a parity target, the image is — the oracle renders from these arrays the very image the HIP pipelines render."""
import numpy as np
import pytest

from conftest import scene_path
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _check_tree(bvh, idx, prims):
    n_nodes, P = bvh.shape[0], prims.shape[0]
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
            assert cnt[n] <= 2 and 0 <= lf[n] and lf[n] + cnt[n] <= P
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
    assert leaves * 4 >= prims.shape[0] and depth < 64              # the reference's traversal stack holds 64 entries
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


def test_gpu_build_small_inputs_and_determinism(gpu_ctx):
    """1, 2, 9 primitives (a single leaf; the smallest trees), duplicates (equal Morton codes), and the same arrays twice."""
    rng = np.random.default_rng(3)
    for n in (1, 2, 8, 9, 100):
        prims = np.zeros((n, 12), np.float32)
        prims[:, 3] = 1.0
        v0 = rng.uniform(-5, 5, (n, 3))
        prims[:, 0:3], prims[:, 4:7], prims[:, 8:11] = v0, v0 + rng.uniform(0, 1, (n, 3)), v0 + rng.uniform(0, 1, (n, 3))
        if n == 100:
            prims[50:] = prims[:50]                                 # exact duplicates: ties in the sort keys
            prims[0, 3], prims[0, 4] = 0.0, 2.5                     # and one sphere
        bvh, idx, ms = gpu_ctx.build_bvh(prims)
        _check_tree(bvh, idx, prims)
        bvh2, idx2, _ = gpu_ctx.build_bvh(prims)
        np.testing.assert_array_equal(bvh.view(np.uint32), bvh2.view(np.uint32))
        np.testing.assert_array_equal(idx, idx2)
        assert (n <= 2) == (bvh.shape[0] == 1)                       # leaves hold <= 2 primitives by default
