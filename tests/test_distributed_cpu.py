"""The N > 1 path on CPU: two processes over gloo run the two partitions of raytracing_amd.parallel.
With no GPU in the process the oracle stands in for the tracer as the per-rank renderer (it takes the
same tile_rank / tile_world / sample_offset options); what is under test is the partition arithmetic
and the collective that combines the ranks' images."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import O, ROOT, WL, bits, oracle_camera, oracle_sd_scene

W, H, SPP = 100, 60, 3


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
    # tile split
    opt = O.make_options(bounce_limit=4, seed=9, threads=2, tile_rank=r, tile_world=w)
    img, _, _ = sc.render(cam, opt, W, H, 1, SPP)
    owner = np.fromfunction(lambda y, x: P.tile_owner(x.astype(int), y.astype(int), W, w), (H, W))
    assert not img[owner != r].any()
    t = torch.from_numpy(img)
    P.combine_tile_split(t, dst=0)
    if r == 0:
        np.save(os.path.join(out_dir, "tile.npy"), t.numpy())
    # sample split
    opt = O.make_options(bounce_limit=4, seed=9, threads=2, sample_offset=P.sample_offset(r, SPP))
    img, _, _ = sc.render(cam, opt, W, H, 1, SPP)
    t = torch.from_numpy(img)
    P.combine_sample_split(t, dst=0)
    if r == 0:
        np.save(os.path.join(out_dir, "sample.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partitions_over_gloo(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 400)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sc = oracle_sd_scene([WL.CYLINDER])
    cam = oracle_camera(WL.DEFAULT_CAMERA)
    full, _, _ = sc.render(cam, O.make_options(bounce_limit=4, seed=9), W, H, 1, SPP)
    assert np.array_equal(bits(np.load(tmp_path / "tile.npy")), bits(full))          # exact
    full2, _, _ = sc.render(cam, O.make_options(bounce_limit=4, seed=9), W, H, 1, world * SPP)
    assert np.allclose(np.load(tmp_path / "sample.npy"), full2, rtol=1e-5, atol=1e-6)  # sum-then-divide vs running mean
