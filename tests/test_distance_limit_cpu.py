"""The argument behind the distance limit of the product's wide walk, checked on the CPU with the oracle alone, with the
constants and leaf classes the product's own upload derives (rt_layout_scene).

Claim: a walk that knows a hit at distance tau may skip every leaf whose box the ray enters at or beyond
(tau + c1 / D + c2 W + c3) (1 + 2 c2 D) -- c1, c2, c3 from the size class of the leaf's triangle, D the largest component
of the ray's direction -- and still return what sd::BVH::IntersectLoop (SimplifiedData.cpp:430-462) returns: a skipped leaf
cannot be a candidate whose triangle test reports a hit at t <= tau, because the fp32 test cannot report a hit further
in front of its own leaf box than that margin, and only such candidates decide the outcome (the smallest (t, position)
is found, and so is every candidate that would force the ordered replay).  ORA_TRAV_LIMITED evaluates exactly that,
nearest leaf box first -- the order in which the limit cuts most -- with the limit taken per LEAF (the product takes the
weaker limit of the largest class below a node), and must agree with the loop bit for bit: on whole renders, on random
rays, on rays that graze, and on rays that start within a micrometre of the surface, where the replay has to run."""
import numpy as np

from helpers import O, R, WL, bits, objparse, oracle_camera

from test_leaves_theorem_cpu import same_hit


def limited_scene(models):
    """The same scene for the oracle and -- through the product's upload layout -- its distance limit."""
    ps, os_ = R.SdScene(), O.SdScene()
    for tris in models:
        ps.add_model(tris)
        os_.add_model(tris, [len(tris)])
    ps.finalize()
    os_.finalize()
    lay = R.device_layout(ps)
    assert lay["has_distance_limit"] and len(lay["wide"]) > 0
    tri_of_slot = lay["tri_pos"][:, 9].view(np.uint32)
    cls = np.full(len(ps.tris()), 3, dtype=np.uint8)
    cls[tri_of_slot] = lay["leaf_class"]
    os_.set_limit(lay["limit_constants"].reshape(9), cls)
    return os_, lay, cls


def test_constants_cover_every_regular_leaf():
    """The class constants are maxima: every leaf of class k has its own c1, c2, c2 W + c3 (twice the bound's) below the
    class's, classes are nested, and the nodes of the wide tree carry the largest class below them."""
    for models in ([objparse.load_obj(WL.DRAGON_2426)[0]], WL.dragon_pair_models()[0:1] + [WL.ground_quad()]):
        os_, lay, cls = limited_scene(models)
        c1, c2x2, c3 = lay["limit_constants"].astype(np.float64)
        assert np.all(np.diff(c1) >= 0) and np.all(np.diff(c2x2) >= 0) and np.all(np.diff(c3) >= 0) and c2x2[2] * 1.25 <= 1.0
        rec = lay["leaf_records"].astype(np.float64)
        a, b = np.abs(rec[:, 3:6]).max(1), np.abs(rec[:, 6:9]).max(1)
        lo, hi = np.concatenate([rec[:, 9:11], rec[:, 12:13]], 1), rec[:, 13:16]
        w = (hi - lo).max(1)
        k = lay["leaf_class"]
        for c in range(3):
            sel = k <= c
            if sel.any():
                assert np.all(2 * (w[sel] + 3e-7 * (a[sel] + b[sel])) <= c1[c] * (1 + 1e-6))
                assert np.all(2 * 340 * a[sel] * b[sel] <= c2x2[c] * (1 + 1e-6))
                assert np.all(340 * a[sel] * b[sel] * w[sel] + 146 * a[sel] * b[sel] * (a[sel] + b[sel]) <= c3[c] * (1 + 1e-6))
        if len(models) == 2:
            assert (k == 3).sum() == 2          # the ground quad's two triangles: far above the dragon's leaves
        # node classes
        wide = lay["wide"]

        def below(ni):
            n_int, n_all = int((wide[ni, 15] >> 24) & 15), int(wide[ni, 15] >> 28)
            base, leaf_base = int(wide[ni, 14]), int(wide[ni, 15] & 0xFFFFFF)
            m = max([int(k[leaf_base + i]) for i in range(n_all - n_int)], default=0)
            for i in range(n_int):
                m = max(m, below(base + i))
            assert int(wide[ni, 13] >> 30) == m
            return m
        below(0)


def test_renders_agree_bit_for_bit_and_leaves_are_skipped():
    quad = WL.ground_quad(y=0.0, x0=-0.5, x1=0.5, z0=-0.5, z1=0.5)
    for models, cam, spp in (([objparse.load_obj(WL.DRAGON_2426)[0]], WL.DRAGON_2426_CAMERA, 2),
                             ([objparse.load_obj(WL.DRAGON_2426)[0], quad], R.Camera(1.0, (0.05, 0.22, 0.45), 2.0, 16.0 / 9.0, (-0.006, 0.06, -0.014)), 1),
                             ([WL.dragon_grid_tris(objparse.load_obj(WL.DRAGON_2426)[0], 3)],           # 3 x 3 dragons, seen from above
                              R.Camera(2.0, (0.244, 0.07 + 0.37, -0.014 + 0.25 + 0.675), 2.0, 16.0 / 9.0, (0.244, 0.07, -0.014 + 0.25)), 1)):
        os_, lay, cls = limited_scene(models)
        a = O.make_options(bounce_limit=6, seed=5, trig=O.TRIG_POLY, traversal=O.TRAV_LOOP)
        b = O.make_options(bounce_limit=6, seed=5, trig=O.TRIG_POLY, traversal=O.TRAV_LIMITED)
        ia, pa, sa = os_.render(oracle_camera(cam), a, 80, 45, 1, spp, want_primary=True)
        O.take_limit_counts()
        ib, pb, sb = os_.render(oracle_camera(cam), b, 80, 45, 1, spp, want_primary=True)
        skipped, met = O.take_limit_counts()
        assert np.array_equal(pa, pb) and np.array_equal(bits(ia), bits(ib))
        assert sa.rays == sb.rays and sa.tri_hits == sb.tri_hits
        assert skipped > 0, (skipped, met)      # the limit is not a dead letter (coarse mesh: large leaves, wide margins)
    # the finer mesh (smaller leaves, smaller margins): a fifth of the leaves whose box a ray meets lie beyond its limit
    os_, lay, cls = limited_scene([objparse.load_obj(WL.DRAGON_18520)[0]])
    cam = oracle_camera(WL.DRAGON_18520_CAMERA)
    ia, pa, sa = os_.render(cam, O.make_options(bounce_limit=6, seed=5, trig=O.TRIG_POLY, traversal=O.TRAV_LOOP, threads=8), 80, 45, 1, 1, want_primary=True)
    O.take_limit_counts()
    ib, pb, sb = os_.render(cam, O.make_options(bounce_limit=6, seed=5, trig=O.TRIG_POLY, traversal=O.TRAV_LIMITED, threads=8), 80, 45, 1, 1, want_primary=True)
    skipped, met = O.take_limit_counts()
    assert np.array_equal(pa, pb) and np.array_equal(bits(ia), bits(ib)) and skipped > 0.15 * met, (skipped, met)


def test_random_grazing_and_surface_rays():
    tris, _ = objparse.load_obj(WL.DRAGON_2426)
    os_, lay, cls = limited_scene([tris])
    t = os_.tris()
    rng = np.random.default_rng(3)
    lo, hi = np.array([-0.12, 0.0, -0.07], np.float32), np.array([0.11, 0.16, 0.04], np.float32)
    O.take_leaf_replays()
    O.take_limit_counts()
    n_hit = 0
    for _ in range(1500):        # rays between random points of the model's neighbourhood, unit and non-unit directions
        o = rng.uniform(lo, hi).astype(np.float32)
        d = (rng.uniform(lo, hi).astype(np.float32) - o)
        if rng.random() < 0.5:
            d = (d / np.linalg.norm(d)).astype(np.float32)
        a, b = os_.intersect(o, d, O.TRAV_LOOP), os_.intersect(o, d, O.TRAV_LIMITED)
        assert same_hit(a, b)
        n_hit += a.hit
    assert n_hit > 300
    # rays that graze: along a triangle's plane, from outside the model, through the triangle's neighbourhood -- the
    # determinant of the triangle test is tiny, its hit distance at its least accurate
    for i in rng.integers(0, len(t), 3000):
        p = t["pos"][i].astype(np.float64)
        n = np.cross(p[1] - p[0], p[2] - p[0])
        n /= np.linalg.norm(n) + 1e-30
        inplane = np.cross(n, rng.normal(0, 1, 3))
        inplane /= np.linalg.norm(inplane) + 1e-30
        d = (inplane + n * rng.normal(0, 1) * 10.0 ** rng.uniform(-7, -2)).astype(np.float32)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        c = p.mean(0) + (p[rng.integers(0, 3)] - p.mean(0)) * rng.uniform(0, 1.2)
        o = (c - d.astype(np.float64) * rng.uniform(0.01, 0.3)).astype(np.float32)
        a, b = os_.intersect(o, d, O.TRAV_LOOP), os_.intersect(o, d, O.TRAV_LIMITED)
        assert same_hit(a, b)
    # rays that start a hair in front of the surface (the winning hit lies in front of its own leaf box: the replay decides)
    for i in rng.integers(0, len(t), 2000):
        p = t["pos"][i].astype(np.float32)
        n = np.cross(p[1] - p[0], p[2] - p[0])
        n = (n / (np.linalg.norm(n) + 1e-30)).astype(np.float32)
        w = rng.dirichlet([0.3, 0.3, 0.3]).astype(np.float32)
        c = (w[0] * p[0] + w[1] * p[1] + w[2] * p[2]).astype(np.float32)
        d = rng.normal(0, 1, 3).astype(np.float32)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        if np.dot(d, n) > 0:
            d = -d
        o = (c - d * np.float32(rng.uniform(1.5e-7, 2e-6))).astype(np.float32)
        a, b = os_.intersect(o, d, O.TRAV_LOOP), os_.intersect(o, d, O.TRAV_LIMITED)
        assert same_hit(a, b)
    skipped, met = O.take_limit_counts()
    assert O.take_leaf_replays() > 50 and skipped > 0
