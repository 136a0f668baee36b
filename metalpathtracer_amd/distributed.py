"""Multi-GPU driver: one process per GPU, pixel tiles sharded across ranks, one RCCL reduce of the HDR sum.

The path shards naturally: every (pixel, sample) is independent and the RNG is counter-based
on (pixel, sample, bounce), so rank r renders the 8x8 tiles t with t % world == r at full spp into a
zero-initialised full-size RGBA32F sum buffer and the only exchange is ONE reduce(sum) of that buffer to
rank 0 (33.2 MB at 1080p, 132.7 MB at 4K; over xGMI).  Each pixel is owned by exactly one rank, so the
reduced image is bit-identical to the 1-GPU image.  `torch.distributed` is plumbing only (backend "nccl" is
RCCL on ROCm; "gloo" on CPU for the tests).
"""
import os

import numpy as np

TILE = 8  # pixels; must match the kernel's 8x8 wave tile


def tile_owner_mask(width, height, rank, world):
    """Boolean [H, W] mask of the pixels rank `rank` owns (tile t = ty * tiles_x + tx, owner = t % world)."""
    tiles_x = (width + TILE - 1) // TILE
    ty, tx = np.divmod(np.arange(((height + TILE - 1) // TILE) * tiles_x), tiles_x)
    owner = (ty * tiles_x + tx) % world
    grid = owner.reshape(-1, tiles_x)
    full = np.repeat(np.repeat(grid, TILE, 0), TILE, 1)[:height, :width]
    return full == rank


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            import torch
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            import torch
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def reduce_framebuffer(sum_tensor, dst=0):
    """In-place reduce(sum) of the [H, W, 4] float32 HDR sum to rank `dst` (no-op for a single process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if sum_tensor.is_cuda and dist.get_backend() == "gloo":
            # functional-test path only (two ranks sharing one GPU cannot use RCCL): stage through the host
            host = sum_tensor.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            sum_tensor.copy_(host)
        else:
            dist.reduce(sum_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return sum_tensor


def render_sharded(render_shard, width, height, rank, world, device="cpu"):
    """Generic sharded render: `render_shard(rank, world) -> [H, W, 4] float32 array/tensor` holding this
    rank's tiles (zeros elsewhere); returns the reduced tensor (valid on rank 0)."""
    import torch
    part = render_shard(rank, world)
    t = part if isinstance(part, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(part))
    t = t.to(device)
    assert tuple(t.shape) == (height, width, 4) and t.dtype == torch.float32
    return reduce_framebuffer(t)
