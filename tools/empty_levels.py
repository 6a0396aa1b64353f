#!/usr/bin/env python3
"""Times a frame whose camera looks away from the scene: every bounce level after the first has no rays, so the
per-launch durations in a kernel trace show the fixed cost of a level."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R  # noqa: E402
from raytracing_amd import workloads as WL  # noqa: E402

wl = WL.make_workload("dragon18520_1080p")
t = R.Tracer(0)
t.upload(wl.scene)
t.set_camera(R.Camera(1.0, (0.0, 0.10, -0.55), 2.0, 16.0 / 9.0, (0.0, 0.10, 10.0)))
t.set_options(bounce_limit=8, seed=1, jitter=0.001, traversal=R.TRAV_FAST)
for i in range(3):
    _, _, st = t.trace(1920, 1080, 1, 16, want_image=False)
    print(f"rays={st.rays} kernel_ms={st.kernel_ms:.3f}", flush=True)
t.close()
