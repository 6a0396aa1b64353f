#!/usr/bin/env python3
"""At the largest configuration (3840x2160, chunks of up to 192 Mi items, so several chunks and rounds that stay large): the
default walk against the plain one, bit for bit, and the ray count.  usage: fullsize_check.py [spp=48]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import raytracing_amd as R
from raytracing_amd import workloads as WL

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 48
wl = WL.make_workload("dragon18520_4k")
t = R.Tracer(0)
t.upload(wl.scene)
t.set_camera(wl.camera)
out = {}
for trav in (R.TRAV_FAST, R.TRAV_FAST_INLINE):
    t.set_options(bounce_limit=wl.bounce_limit, seed=wl.seed, jitter=wl.jitter, traversal=trav)
    img, prim, st = t.trace(wl.width, wl.height, 1, spp, want_primary=True)
    out[trav] = (img.view(np.uint32).copy(), prim.copy(), st.rays, st.kernel_ms_sum)
    print("trav", trav, "rays", st.rays, "kernel ms", round(st.kernel_ms_sum, 1), flush=True)
a, b = out[R.TRAV_FAST], out[R.TRAV_FAST_INLINE]
assert a[2] == b[2] and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
print("bit-identical:", wl.width, "x", wl.height, spp, "spp")
