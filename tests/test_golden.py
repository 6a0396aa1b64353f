"""The oracle against the committed golden fixtures (CPU only).  Regenerate with tests/golden/make_golden.py."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, O, R, WL, bits, oracle_camera, oracle_old_scene, oracle_sd_scene

with open(os.path.join(GOLDEN, "golden.json")) as fh:
    META = json.load(fh)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name,path", [("dragon2426", WL.DRAGON_2426), ("dragon18520", WL.DRAGON_18520),
                                       ("cylinder", WL.CYLINDER)])
def test_bvh_node_arrays(name, path):
    sc = oracle_sd_scene([path])
    nodes = sc.nodes()
    m = META[name]
    assert sc.counts() == (m["n_nodes"], m["n_tris"], m["root"])
    assert nodes[:16].tobytes().hex() == m["first16"] and nodes[-16:].tobytes().hex() == m["last16"]
    assert sha(nodes) == m["nodes_sha256"]


def test_dragon2426_primary_ids():
    sc = oracle_sd_scene([WL.DRAGON_2426])
    m = META["dragon2426"]["primary_400x225"]
    opt = O.make_options(bounce_limit=8, jitter=0.0, trig=O.TRIG_POLY)
    _, prim, st = sc.render(oracle_camera(WL.DRAGON_2426_CAMERA), opt, 400, 225, 1, 1, want_primary=True)
    gold = np.load(os.path.join(GOLDEN, "dragon2426_primary_400x225.npz"))["prim"]
    assert np.array_equal(prim, gold)
    assert (st.rays, st.box_tests, st.tri_tests, st.tri_hits) == (m["rays"], m["box_tests"], m["tri_tests"], m["tri_hits"])


def test_dragon2426_shaded():
    sc = oracle_sd_scene([WL.DRAGON_2426])
    opt = O.make_options(bounce_limit=8, seed=7, jitter=0.001, trig=O.TRIG_POLY)
    img, _, st = sc.render(oracle_camera(WL.DRAGON_2426_CAMERA), opt, 96, 54, 1, 8)
    gold = np.load(os.path.join(GOLDEN, "dragon2426_shaded_96x54_8spp.npy"))
    assert np.array_equal(bits(img), bits(gold))


def test_dragon18520_small_render():
    sc = oracle_sd_scene([WL.DRAGON_18520])
    opt = O.make_options(bounce_limit=8, seed=1, jitter=0.001, trig=O.TRIG_POLY)
    img, prim, st = sc.render(oracle_camera(WL.DRAGON_18520_CAMERA), opt, 96, 54, 1, 1, want_primary=True)
    g = np.load(os.path.join(GOLDEN, "dragon18520_96x54_1spp.npz"))
    m = META["dragon18520"]["render_96x54_1spp"]
    assert np.array_equal(prim, g["prim"]) and np.array_equal(bits(img), bits(g["img"]))
    assert (st.rays, st.box_tests, st.tri_tests) == (m["rays"], m["box_tests"], m["tri_tests"])


def test_small_spheres_shaded_8spp():
    sc = oracle_old_scene(WL.small_sphere_scene())
    cam = R.Camera(3.19, (13.0, 2.0, 3.0), 2.0, 16.0 / 9.0, (0.0, 0.5, 0.0))
    opt = O.make_options(bounce_limit=8, seed=1, jitter=0.001, trig=O.TRIG_POLY, accum=O.ACCUM_FORWARD)
    img, prim, st = sc.render(oracle_camera(cam), opt, 64, 36, 1, 8, want_primary=True)
    assert np.array_equal(bits(img), bits(np.load(os.path.join(GOLDEN, "spheres_small_64x36_8spp.npy"))))
    m = META["spheres_small_8spp"]
    assert (st.rays, st.box_tests, st.sph_tests, sha(prim)) == (m["rays"], m["box_tests"], m["sph_tests"], m["primary_sha256"])
