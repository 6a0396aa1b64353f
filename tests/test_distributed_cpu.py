"""The N > 1 path on CPU: two processes over gloo run raytracing_amd.parallel.render -- the function bench.py runs per
rank -- for both partitions, twice in a row.  With no GPU in the process the oracle stands in for the tracer as the
per-rank renderer (it takes the same tile_rank / tile_world / sample_offset options and continues a running mean the
same way); what is under test is the partition arithmetic, the progressive accumulation across passes and the
collective that combines the ranks' images, including the state the first render leaves behind for the second."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import O, ROOT, WL, bits, oracle_camera, oracle_sd_scene

W, H, SPP, STEPS = 100, 60, 3, 2   # three samples per pass over two ranks: unequal shares (2 + 1)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from raytracing_amd import parallel as P
    r, w, _ = P.init_from_env("gloo")
    assert (r, w) == (rank, world)
    sc = oracle_sd_scene([WL.CYLINDER])
    cam = oracle_camera(WL.DEFAULT_CAMERA)
    for split in ("tile", "sample"):
        per_rank = P.samples_of_rank(SPP, r, w)
        share = P.share_options(split, r, w, spp=SPP, steps=STEPS)
        if split == "sample":   # rank 0 draws the global sample indices 1 .. 4, rank 1 the indices 5, 6
            assert share == {"sample_offset": (0, STEPS * 2)[r]} and per_rank == (2, 1)[r]
        opt = O.make_options(bounce_limit=4, seed=9, threads=2, **share)
        t = torch.zeros((H, W, 4), dtype=torch.float32)
        n = SPP if split == "tile" else per_rank
        weight = P.sample_weight(r, w, SPP) if split == "sample" else None

        def trace_pass(i):
            sc.render(cam, opt, W, H, 1 + i * n, n, image=t.numpy())

        for attempt in range(2):   # the second render starts from what the first one's reduce left in `t`
            P.render(trace_pass, lambda: None, t, STEPS, split, w, weight=weight)
            if r == 0:
                np.save(os.path.join(out_dir, f"{split}{attempt}.npy"), t.numpy())
        if split == "tile":
            own, _, _ = sc.render(cam, opt, W, H, 1, 1)
            owner = np.fromfunction(lambda y, x: P.tile_owner(x.astype(int), y.astype(int), W, w), (H, W))
            assert not own[owner != r].any() and own[owner == r].any()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partitions_over_gloo(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 400)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sc = oracle_sd_scene([WL.CYLINDER])
    cam = oracle_camera(WL.DEFAULT_CAMERA)
    full, _, _ = sc.render(cam, O.make_options(bounce_limit=4, seed=9), W, H, 1, STEPS * SPP)
    for attempt in range(2):
        assert np.array_equal(bits(np.load(tmp_path / f"tile{attempt}.npy")), bits(full))          # exact, both times
        # weighted sum of two running means over disjoint parts (4 + 2) of the same sample indices vs one running mean
        assert np.allclose(np.load(tmp_path / f"sample{attempt}.npy"), full, rtol=1e-5, atol=1e-6)


def _single_worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RT_FORCE_DIST="1")
    from raytracing_amd import parallel as P
    r, w, _ = P.init_from_env("gloo")
    assert (r, w) == (0, 1) and dist.is_initialized()   # a group of one: the collectives run, and change nothing
    sc = oracle_sd_scene([WL.CYLINDER])
    cam = oracle_camera(WL.DEFAULT_CAMERA)
    for split in ("tile", "sample"):
        assert P.share_options(split, 0, 1) == {}
        opt = O.make_options(bounce_limit=4, seed=9, threads=2)
        t = torch.zeros((H, W, 4), dtype=torch.float32)
        P.render(lambda i: sc.render(cam, opt, W, H, 1 + i * SPP, SPP, image=t.numpy()), lambda: None, t, STEPS, split, 1)
        np.save(os.path.join(out_dir, f"one_{split}.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_taken_through_the_collectives(tmp_path):
    """RT_FORCE_DIST=1 (the single-GPU rehearsal of the multi-rank bench): the reduce of a group of one leaves the
    image bit for bit what a render without any process group gives."""
    mp.spawn(_single_worker, args=(29900 + (os.getpid() % 90), str(tmp_path)), nprocs=1, join=True)
    sc = oracle_sd_scene([WL.CYLINDER])
    full, _, _ = sc.render(oracle_camera(WL.DEFAULT_CAMERA), O.make_options(bounce_limit=4, seed=9), W, H, 1, STEPS * SPP)
    for split in ("tile", "sample"):
        assert np.array_equal(bits(np.load(tmp_path / f"one_{split}.npy")), bits(full))
