#!/usr/bin/env python3
"""Compact-tree walk against the 32-byte-record walk: same image (sha256 of the mean) and counters, kernel ms of each.
Runs itself once per setting of RT_COMPACT (the library reads its tuning knobs once per process)."""
import hashlib
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [("dragon18520_1080p", 0, 0, 64, 0), ("dragon18520_full_ground_1080p", 0, 0, 16, 11), ("dragon2426_1080p", 0, 0, 16, 0),
         ("dragon18520_bare_4k", 0, 0, 8, 11), ("dragon18520_bare_4k", 0, 0, 8, 0),
         ("dragon_field_k8_1080p", 0, 0, 16, 11), ("dragon_field_k8_1080p", 0, 0, 16, 0), ("dragon_field_k23_1080p", 0, 0, 8, 11),
         ("dragon_field_k23_1080p", 0, 0, 8, 0), ("dragon18520_full_ground_1080p", 0, 0, 16, 7),
         ("dragon_field_k8_1080p", 0, 0, 16, 7), ("dragon_field_k23_1080p", 0, 0, 8, 7), ("dragon18520_1080p", 0, 0, 64, 7)]
if os.environ.get("AB_CASES"):
    CASES = [CASES[int(i)] for i in os.environ["AB_CASES"].split(",")]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import raytracing_amd as R
    from raytracing_amd import workloads as WL
    for name, W, H, spp, trav in CASES:
        wl = WL.make_workload(name)
        t = R.Tracer(0)
        t.upload(wl.scene)
        t.set_camera(wl.camera)
        t.set_options(bounce_limit=wl.bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter, traversal=trav)
        best = 1e9
        for _ in range(3):
            img, _, st = t.trace(W or wl.width, H or wl.height, 1, spp, want_image=True)
            best = min(best, st.kernel_ms)
        t.set_options(bounce_limit=wl.bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter, traversal=trav, collect_stats=True)
        _, _, s2 = t.trace(W or wl.width, H or wl.height, 1, spp, want_image=False)
        print(f"{name} trav={trav} spp={spp}: {best:.3f} ms rays={st.rays} sha={hashlib.sha256(img.tobytes()).hexdigest()[:16]} "
              f"box={s2.box_tests} tri={s2.tri_tests} hits={s2.tri_hits} replays={s2.replays}", flush=True)
        t.close()
else:
    for c in ("0", "1", "0", "1"):
        print(f"--- RT_COMPACT={c}", flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, RT_COMPACT=c), timeout=400)
        if r.returncode != 0:
            sys.exit(r.returncode)
