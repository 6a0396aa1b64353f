"""The argument behind the product's fast walk, checked on the CPU with the oracle alone.

Claim: what sd::BVH::IntersectLoop (SimplifiedData.cpp:430-462) returns is a function of the tree's LEAVES -- the
interior boxes only prune -- namely the candidate (leaf box met with closest = +inf, triangle hit) with the smallest
(t, position in the visiting order), provided its box entry distance m is <= t; otherwise the result of playing the
acceptance rule "closest > m and t < closest" over the candidates in visiting order.  ORA_TRAV_LEAVES evaluates exactly
that, visiting the leaves in reverse storage order, and must agree with the loop bit for bit: on whole renders, on
random rays, and on rays that start within a micrometre of the surface, where the hit distance falls below the slab
test's lower bound, the winning hit lies in front of its own leaf box, and the replay has to run.
"""
import numpy as np

from helpers import O, WL, bits, oracle_camera, oracle_sd_scene


def same_hit(a, b):
    return (a.hit, a.prim, np.float32(a.t).tobytes(), bytes(a.pos), bytes(a.normal)) == \
           (b.hit, b.prim, np.float32(b.t).tobytes(), bytes(b.pos), bytes(b.normal))


def test_renders_agree_bit_for_bit():
    for path, cam in ((WL.DRAGON_2426, WL.DRAGON_2426_CAMERA), (WL.CYLINDER, WL.DEFAULT_CAMERA)):
        sc = oracle_sd_scene([path])
        a = O.make_options(bounce_limit=8, seed=5, trig=O.TRIG_POLY, traversal=O.TRAV_LOOP)
        b = O.make_options(bounce_limit=8, seed=5, trig=O.TRIG_POLY, traversal=O.TRAV_LEAVES)
        ia, pa, sa = sc.render(oracle_camera(cam), a, 96, 54, 1, 3, want_primary=True)
        O.take_leaf_replays()
        ib, pb, sb = sc.render(oracle_camera(cam), b, 96, 54, 1, 3, want_primary=True)
        assert np.array_equal(pa, pb) and np.array_equal(bits(ia), bits(ib))
        assert sa.rays == sb.rays and sa.tri_hits == sb.tri_hits


def test_quirk_tree_dragon18520():
    """The tree whose interior boxes are stretched to z = 0 (FLT_MIN as initial pMax): same hits from the leaves."""
    sc = oracle_sd_scene([WL.DRAGON_18520])
    a = O.make_options(bounce_limit=3, seed=2, trig=O.TRIG_POLY, traversal=O.TRAV_LOOP)
    b = O.make_options(bounce_limit=3, seed=2, trig=O.TRIG_POLY, traversal=O.TRAV_LEAVES)
    cam = oracle_camera(WL.DRAGON_18520_CAMERA)
    ia, pa, _ = sc.render(cam, a, 64, 36, 1, 1, want_primary=True)
    ib, pb, _ = sc.render(cam, b, 64, 36, 1, 1, want_primary=True)
    assert (pa != O.MISS).sum() > 50
    assert np.array_equal(pa, pb) and np.array_equal(bits(ia), bits(ib))


def test_random_and_grazing_rays_including_replays():
    sc = oracle_sd_scene([WL.DRAGON_2426])
    tris = sc.tris()
    rng = np.random.default_rng(11)
    O.take_leaf_replays()
    n_hit = 0
    # rays between random points of the model's neighbourhood
    lo, hi = np.array([-0.12, 0.0, -0.07], np.float32), np.array([0.11, 0.16, 0.04], np.float32)
    for _ in range(1500):
        o = rng.uniform(lo, hi).astype(np.float32)
        d = (rng.uniform(lo, hi).astype(np.float32) - o)
        a, b = sc.intersect(o, d, O.TRAV_LOOP), sc.intersect(o, d, O.TRAV_LEAVES)
        assert same_hit(a, b)
        n_hit += a.hit
    assert n_hit > 300
    random_replays = O.take_leaf_replays()
    # rays that start a hair in front of the surface: the hit distance lies between the triangle test's 1e-7 and the slab
    # test's 1e-6 lower bound, so the winning hit is in front of its own leaf box's entry distance (m = 1e-6 > t) and
    # the ordered replay has to decide; next to an edge or a vertex several such hits compete and the visiting order matters
    for i in rng.integers(0, len(tris), 3000):
        p = tris["pos"][i].astype(np.float32)
        n = np.cross(p[1] - p[0], p[2] - p[0])
        n = (n / (np.linalg.norm(n) + 1e-30)).astype(np.float32)
        w = rng.dirichlet([0.3, 0.3, 0.3]).astype(np.float32)          # often close to an edge or a vertex
        c = (w[0] * p[0] + w[1] * p[1] + w[2] * p[2]).astype(np.float32)
        d = rng.normal(0, 1, 3).astype(np.float32)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        if np.dot(d, n) > 0:
            d = -d
        o = (c - d * np.float32(rng.uniform(1.5e-7, 2e-6))).astype(np.float32)
        a, b = sc.intersect(o, d, O.TRAV_LOOP), sc.intersect(o, d, O.TRAV_LEAVES)
        assert same_hit(a, b)
    close_replays = O.take_leaf_replays()
    # the replay is not a dead branch
    assert close_replays > 50, (random_replays, close_replays)
