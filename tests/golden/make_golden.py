#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the CPU oracle.

The reference holds no golden vectors for this path (parity unpinned), so these fixtures pin the
ORACLE against regressions and give the GPU tests committed expected outputs: BVH node-array digests,
primary-hit id images, visit counters and small shaded images.  Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import O, R, WL, oracle_camera, oracle_old_scene, oracle_sd_scene  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    meta = {}
    for name, path, cam in (("dragon2426", WL.DRAGON_2426, WL.DRAGON_2426_CAMERA),
                            ("dragon18520", WL.DRAGON_18520, WL.DRAGON_18520_CAMERA),
                            ("cylinder", WL.CYLINDER, WL.DEFAULT_CAMERA)):
        sc = oracle_sd_scene([path])
        nodes = sc.nodes()
        n, t, root = sc.counts()
        meta[name] = {"n_nodes": n, "n_tris": t, "root": root, "nodes_sha256": sha(nodes),
                      "first16": nodes[:16].tobytes().hex(), "last16": nodes[-16:].tobytes().hex()}
        if name == "dragon2426":
            opt = O.make_options(bounce_limit=8, jitter=0.0, trig=O.TRIG_POLY)
            _, prim, st = sc.render(oracle_camera(cam), opt, 400, 225, 1, 1, want_primary=True)
            np.savez_compressed(os.path.join(HERE, "dragon2426_primary_400x225.npz"), prim=prim)
            meta[name]["primary_400x225"] = {"rays": st.rays, "box_tests": st.box_tests, "tri_tests": st.tri_tests,
                                             "tri_hits": st.tri_hits, "sha256": sha(prim)}
            opt1 = O.make_options(bounce_limit=1, jitter=0.0, trig=O.TRIG_POLY)
            _, prim, st = sc.render(oracle_camera(cam), opt1, 1920, 1080, 1, 1, want_primary=True)
            meta[name]["primary_1080p"] = {"sha256": sha(prim), "hits": int((prim != O.MISS).sum()),
                                           "box_tests": st.box_tests, "tri_tests": st.tri_tests}
            opt8 = O.make_options(bounce_limit=8, seed=7, jitter=0.001, trig=O.TRIG_POLY)
            img, _, st = sc.render(oracle_camera(cam), opt8, 96, 54, 1, 8)
            np.save(os.path.join(HERE, "dragon2426_shaded_96x54_8spp.npy"), img)
            meta[name]["shaded_96x54_8spp"] = {"rays": st.rays, "box_tests": st.box_tests, "tri_tests": st.tri_tests}
        if name == "dragon18520":
            opt = O.make_options(bounce_limit=8, seed=1, jitter=0.001, trig=O.TRIG_POLY)
            img, prim, st = sc.render(oracle_camera(cam), opt, 96, 54, 1, 1, want_primary=True)
            np.savez_compressed(os.path.join(HERE, "dragon18520_96x54_1spp.npz"), img=img, prim=prim)
            meta[name]["render_96x54_1spp"] = {"rays": st.rays, "box_tests": st.box_tests, "tri_tests": st.tri_tests}
    sp = WL.small_sphere_scene()
    sc = oracle_old_scene(sp)
    cam = R.Camera(3.19, (13.0, 2.0, 3.0), 2.0, 16.0 / 9.0, (0.0, 0.5, 0.0))
    for spp in (8, 256):
        opt = O.make_options(bounce_limit=8, seed=1, jitter=0.001, trig=O.TRIG_POLY, accum=O.ACCUM_FORWARD)
        img, prim, st = sc.render(oracle_camera(cam), opt, 64, 36, 1, spp, want_primary=True)
        np.save(os.path.join(HERE, f"spheres_small_64x36_{spp}spp.npy"), img)
        meta[f"spheres_small_{spp}spp"] = {"rays": st.rays, "box_tests": st.box_tests, "sph_tests": st.sph_tests,
                                           "primary_sha256": sha(prim)}
    with open(os.path.join(HERE, "golden.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
