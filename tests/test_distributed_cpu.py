"""N > 1 path on CPU: two `gloo` ranks shard the 8x8 tiles, each renders its shard, ONE reduce(sum) of the
framebuffer lands the full image on rank 0 — bit-identical to the single-rank image.  On the GPU box the same
code runs with backend nccl (= RCCL) and mpt_render(shard_rank, shard_count); here the oracle stands in for
the renderer (as the checker's renderer, masked to the rank's tiles) so the sharding arithmetic and the
collective plumbing are what is under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    torch.set_num_threads(1)
    from metalpathtracer_amd import distributed as D
    from oracle import binding as ob
    r, w, _ = D.init_from_env(backend="gloo")
    W, H, spp = 100, 60, 2
    sc = ob.OracleScene()
    assert sc.load_xml(os.path.join(ROOT, "assets", "scene.xml")) == 0
    sc.build_bvh()
    buf = sc.buffers()
    u = ob.make_uniforms(W, H, sc.prim_count, sc.triangle_count)

    def shard(rank_, world_):
        img, _ = ob.render(u, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=spp)
        return img * D.tile_owner_mask(W, H, rank_, world_)[..., None]

    t = D.render_sharded(shard, W, H, r, w)
    if r == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), t.numpy())
        full, _ = ob.render(u, buf, rng_mode=ob.RNG_PHILOX, max_depth=8, accumulate=1, sample_count=spp)
        np.save(os.path.join(out_dir, "full.npy"), full)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_reduce(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    red = np.load(tmp_path / "reduced.npy")
    full = np.load(tmp_path / "full.npy")
    np.testing.assert_array_equal(red.view(np.uint32), full.view(np.uint32))


@pytest.mark.parametrize("W,H,world", [(1920, 1080, 8), (100, 60, 2), (101, 67, 3), (8, 8, 4), (3840, 2160, 8)])
def test_tile_masks_partition_the_image(W, H, world):
    from metalpathtracer_amd import distributed as D
    total = np.zeros((H, W), np.int32)
    sizes = []
    for r in range(world):
        m = D.tile_owner_mask(W, H, r, world)
        total += m
        sizes.append(int(m.sum()))
    assert (total == 1).all()
    if W * H >= 64 * 64 * world:
        assert max(sizes) - min(sizes) <= 0.02 * W * H / world + 64 * 8   # interleaved tiles balance pixel counts
