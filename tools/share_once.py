#!/usr/bin/env python3
"""Three trace calls of rank 0's tile share of the headline frame: share_once.py <world> <spp>; prints world, spp, kernel ms.
Run under rocprofv3 --kernel-trace --stats to see what a share's call is made of, launch by launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R
from raytracing_amd import workloads as WL
wl = WL.make_workload("dragon18520_1080p")
t = R.Tracer(0); t.upload(wl.scene); t.set_camera(wl.camera)
world = int(sys.argv[1]); spp = int(sys.argv[2])
t.set_options(bounce_limit=wl.bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter, tile_rank=0, tile_world=world)
for i in range(3):
    _, _, st = t.trace(wl.width, wl.height, 1, spp, want_image=False)
    print(world, spp, st.kernel_ms, flush=True)
t.close()
