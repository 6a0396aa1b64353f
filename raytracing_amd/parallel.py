"""Multi-GPU partitioning of one render, one process per GPU (torch.distributed; the "nccl" backend is RCCL over xGMI
on ROCm), scene replicated on every GPU, no collective inside the trace itself.  (Several GPUs inside ONE process are
rt_create_multi's job, include/rt_hip.h; the two compose.)

Two partitions of the per-pixel loop, both embarrassingly parallel (the reference already splits image rows over 16
threads with no communication, Renderer/TraceMethods.cpp:61-72):

* tile split    -- rank r renders the 8x8 tiles k with k % world == r (rt_options.tile_rank/tile_world); pixels of
                   other ranks are zero, so ONE reduce(SUM) to rank 0 assembles the image and is exact (x + 0 == x).
* sample split  -- rank r renders its own share of the samples of every pixel, with the random streams of its own
                   sample indices (rt_options.sample_offset); ONE reduce(SUM) of the per-rank means, each weighted by
                   its share of the samples, gives the mean over all samples.  That is a sum-then-divide, not the reference's sequential running
                   mean, so it agrees with a single-GPU render to rounding, not bit for bit.

A render is `n_steps` trace passes accumulated progressively into the rank's image (the reference's sampleCount
protocol, Renderer/TracerImpl.cpp:9-12) followed by that one reduce.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Returns (rank, world, local_rank); initialises the process group when WORLD_SIZE > 1 -- or when RT_FORCE_DIST=1,
    which takes a single rank through the very collectives of a multi-rank run (RCCL init, barrier, reduce, all-reduce):
    the rehearsal a box with one GPU allows."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("RT_FORCE_DIST") == "1") and not dist.is_initialized():
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


def samples_of_rank(spp, rank, world):
    """Samples of a pass of `spp` samples per pixel that rank `rank` renders in a sample split: the first spp % world
    ranks take one more than the others."""
    return spp // world + (1 if rank < spp % world else 0)


def sample_offset(rank, world, spp, steps=1):
    """RNG sample-index offset of `rank` when a render is `steps` passes of `spp` samples per pixel shared by `world` ranks:
    every rank numbers its own samples 1 .. steps * samples_of_rank(spp, rank, world) and draws the global indices that
    follow those of the ranks before it (shares may differ by one per pass, so the offset is the sum of the preceding
    ranks' totals, not rank times one's own)."""
    return steps * sum(samples_of_rank(spp, q, world) for q in range(rank))


def sample_weight(rank, world, spp):
    """This rank's share of the samples, the weight of its mean in the combined image."""
    return samples_of_rank(spp, rank, world) / float(spp)


def tile_owner(x, y, width, world):
    """Rank that renders pixel (x, y) in a tile split."""
    tiles_x = (width + 7) // 8
    return ((y // 8) * tiles_x + (x // 8)) % world


def share_options(split, rank, world, spp=0, steps=1):
    """The rt_options fields that give `rank` its share: tile_rank / tile_world, or the sample offset of a render of
    `steps` passes of `spp` samples per pixel (all ranks together)."""
    if world <= 1:
        return {}
    if split == "tile":
        return {"tile_rank": rank, "tile_world": world}
    return {"sample_offset": sample_offset(rank, world, spp, steps)}


def _reduce_sum(image, dst):
    """reduce(SUM) onto rank `dst`, in place on `dst`.  Over RCCL the other ranks' tensors are left as they were; gloo
    (CPU tests, rehearsals with several ranks on one GPU) uses them as scratch, so they hold no image afterwards.

    gloo with a DEVICE tensor: the host waits for the GPU first.  On this ROCm / torch build a gloo reduce of a device
    tensor that is entered while kernels are still in flight never returns from the second such call on, when two ranks
    share one GPU -- with plain torch kernels and no tracer in the process just as with the tracer
    (tools/gloo_probe.py, modes busy / many against many_sync).  RCCL orders itself behind the current stream and needs
    no host wait."""
    if image.is_cuda and dist.get_backend() == "gloo":
        torch.cuda.synchronize(image.device)
    dist.reduce(image, dst=dst, op=dist.ReduceOp.SUM)


def combine_tile_split(image, dst=0):
    """image: this rank's (H, W, 4) float32 tensor, ZERO outside its tiles.  After the call rank `dst` holds the
    assembled image."""
    if dist.is_initialized():
        _reduce_sum(image, dst)
    return image


def combine_sample_split(image, dst=0, weight=None):
    """image: this rank's mean over its own samples.  After the call rank `dst` holds the mean over all ranks'
    samples.  weight = this rank's share of the samples (None: equal shares, 1 / world)."""
    if dist.is_initialized():
        world = dist.get_world_size()
        if weight is not None:
            image *= weight
        _reduce_sum(image, dst)
        if weight is None and dist.get_rank() == dst:
            image /= world
    return image


def render(trace_pass, join, image, n_steps, split, world, weight=None):
    """One render on this rank: `n_steps` passes accumulated into `image`, then the one reduce.

    trace_pass(i): enqueue pass i (samples of step i) into `image` as a running mean, first pass restarting it;
    join(): order the image after everything enqueued (a stream-level wait is enough).
    weight (sample split): this rank's share of the samples, sample_weight(); None = equal shares.
    In a tile split a pass writes this rank's tiles only, and on rank `dst` the previous render's reduce left the other
    ranks' pixels in `image`: the image is cleared first, so that "zero outside its tiles" holds for every render."""
    if split == "tile" and world > 1:
        image.zero_()
    for i in range(n_steps):
        trace_pass(i)
    join()
    if split == "tile":
        combine_tile_split(image, dst=0)
    else:
        combine_sample_split(image, dst=0, weight=weight)
    return image
