#!/usr/bin/env python3
"""Runs a few trace launches of one workload through the C ABI (no torch import: quick to start under
rocprofv3).  Prints rays, kernel ms and Mrays/s per launch."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R  # noqa: E402
from raytracing_amd import workloads as WL  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="dragon18520_1080p")
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--height", type=int, default=0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--stats", action="store_true")
ap.add_argument("--bounce", type=int, default=0)
ap.add_argument("--trav", type=int, default=0)
ap.add_argument("--camera", type=int, default=0, help="RT_CAMERA_*: 0 auto, 1 separate camera stage, 2 fused")
ap.add_argument("--image", action="store_true", help="also copy the mean image to the host (PCIe-inclusive total_ms)")
a = ap.parse_args()
wl = WL.make_workload(a.workload)
W, H, spp = a.width or wl.width, a.height or wl.height, a.spp or wl.spp
t = R.Tracer(0)
t.upload(wl.scene)
t.set_camera(wl.camera)
t.set_options(bounce_limit=a.bounce or wl.bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter,
              collect_stats=a.stats, traversal=a.trav, camera_stage=a.camera)
for i in range(a.reps):
    _, _, st = t.trace(W, H, 1, spp, want_image=a.image)
    print(f"{a.workload} {W}x{H} spp={spp} rep={i}: rays={st.rays} kernel_ms={st.kernel_ms:.3f} "
          f"Mrays/s={st.rays / st.kernel_ms / 1e3:.2f} total_ms={st.total_ms:.3f} box={st.box_tests} tri={st.tri_tests} sph={st.sph_tests} replays={st.replays}", flush=True)
t.close()
