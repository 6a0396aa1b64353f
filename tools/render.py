#!/usr/bin/env python3
"""Renders a named workload on the GPU and writes the displayed image (gamma 2.2) as PNG or PPM."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R  # noqa: E402
from raytracing_amd import imageio, workloads as WL  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="dragon18520_1080p")
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--height", type=int, default=0)
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--trav", type=int, default=0)
ap.add_argument("--out", default="gpurun_out/render.png")
a = ap.parse_args()
t0 = time.time()
wl = WL.make_workload(a.workload)
t1 = time.time()
W, H, spp = a.width or wl.width, a.height or wl.height, a.spp or wl.spp
t = R.Tracer(0)
t.upload(wl.scene)
t2 = time.time()
t.set_camera(wl.camera)
t.set_options(bounce_limit=wl.bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter, traversal=a.trav)
img, _, st = t.trace(W, H, 1, spp)
t.close()
(imageio.write_ppm if a.out.endswith(".ppm") else imageio.write_png)(a.out, img)
print(f"{a.workload} {W}x{H} spp={spp}: host scene {t1 - t0:.2f} s, upload {t2 - t1:.2f} s, rays={st.rays} "
      f"kernel_ms={st.kernel_ms:.3f} Mrays/s={st.rays / st.kernel_ms / 1e3:.1f} replays={st.replays} -> {a.out}")
