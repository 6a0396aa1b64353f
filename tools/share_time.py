#!/usr/bin/env python3
"""Kernel time of one rank's share of a step under the tile split and under the sample split, next to the whole step on
one GPU: what strong scaling over N GPUs can reach at best (one GPU plays rank 0 of N)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R  # noqa: E402
from raytracing_amd import workloads as WL  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dragon18520_1080p"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 512
wl = WL.make_workload(name)
t = R.Tracer(0)
t.upload(wl.scene)
t.set_camera(wl.camera)


def run(n, **kw):
    t.set_options(bounce_limit=wl.bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter, **kw)
    best = 1e9
    for _ in range(3):
        _, _, st = t.trace(wl.width, wl.height, 1, n)
        best = min(best, st.kernel_ms)
    return best, st.rays


whole, rays = run(spp)
print(f"{name} {spp} spp whole: {whole:.2f} ms ({rays / whole / 1e3:.0f} Mrays/s)", flush=True)
for world in (2, 4, 8):
    tile, _ = run(spp, tile_rank=0, tile_world=world)
    samp, _ = run(spp // world, sample_offset=0)
    print(f"  N={world}: tile share {tile:.2f} ms -> x{whole / tile:.2f};  sample share {samp:.2f} ms -> x{whole / samp:.2f}", flush=True)
t.close()
