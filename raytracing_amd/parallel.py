"""Multi-GPU partitioning of one render: one process per GPU (torch.distributed; the "nccl" backend is
RCCL over xGMI on ROCm), scene replicated on every GPU, no collective inside the trace itself.

Two partitions of the per-pixel loop, both embarrassingly parallel (the reference already splits image
rows over 16 threads with no communication, Renderer/TraceMethods.cpp:61-72):

* tile split    -- rank r renders the 8x8 tiles k with k % world == r (rt_options.tile_rank/tile_world);
                   pixels of other ranks stay zero, so ONE reduce(SUM) to rank 0 assembles the image and
                   is exact (x + 0 == x).
* sample split  -- rank r renders its own `spp` samples of every pixel with random streams of sample
                   indices r*spp+1 .. (r+1)*spp (rt_options.sample_offset); ONE reduce(SUM) of the
                   per-rank means, divided by world, gives the mean over world*spp samples.  That is
                   a sum-then-divide, not the reference's sequential running mean, so it agrees with a
                   single-GPU render to rounding, not bit for bit.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Returns (rank, world, local_rank); initialises the process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def sample_offset(rank, spp):
    """RNG sample-index offset of `rank` in a sample split with `spp` samples per rank."""
    return rank * spp


def tile_owner(x, y, width, world):
    """Rank that renders pixel (x, y) in a tile split."""
    tiles_x = (width + 7) // 8
    return ((y // 8) * tiles_x + (x // 8)) % world


def _reduce_sum(image, dst):
    """reduce(SUM) onto rank `dst`.  RCCL reduces device tensors in place over xGMI.  gloo (used only to rehearse
    the multi-rank path where RCCL cannot run, e.g. several ranks sharing one GPU) is given a host copy: its
    handling of device tensors proved unreliable here (a second reduce of the same CUDA tensor never returned)."""
    if image.is_cuda and dist.get_backend() == "gloo":
        host = image.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
        image.copy_(host)
    else:
        dist.reduce(image, dst=dst, op=dist.ReduceOp.SUM)


def combine_tile_split(image, dst=0):
    """image: this rank's (H, W, 4) float32 tensor, zero outside its tiles.  After the call rank `dst`
    holds the assembled image."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        _reduce_sum(image, dst)
    return image


def combine_sample_split(image, dst=0):
    """image: this rank's mean over its own samples.  After the call rank `dst` holds the mean over all
    ranks' samples."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        _reduce_sum(image, dst)
        if dist.get_rank() == dst:
            image /= world
    return image
