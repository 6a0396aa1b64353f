"""GPU parity tests for the flat-array ("sd") path, through the C ABI (rt_trace).

Bar: bit-exact.  Primary-hit primitive ids, slab/triangle test counts and every float of the shaded
running-mean image must equal the oracle's (oracle trig = the polynomial cos/sin the kernel uses);
against the oracle's libm cos/sin the stated tolerance is |gpu-cpu| <= 1e-4 + 1e-3*|cpu| on >= 99.9 %
of the pixels.
"""
import numpy as np
import pytest

from helpers import O, R, WL, bits, objparse, oracle_camera, oracle_sd_scene, product_sd_scene, psnr

pytestmark = pytest.mark.gpu

# every traversal variant must give the same bits: the wave-synchronous walk (scalar node fetches), the
# per-lane walk, the first-generation nested-loop kernel (these three perform the reference's own tests, so
# their test counts equal the oracle's too), the fast walk over the device's own hierarchy, and the fast
# walk with every ray forced through its ordered replay
TRAVS = [R.TRAV_WAVE, R.TRAV_LANE, R.TRAV_NESTED, R.TRAV_FAST, R.TRAV_FAST_REPLAY, R.TRAV_FAST_REFILL, R.TRAV_FAST_QUEUED, R.TRAV_FAST_INLINE, R.TRAV_FAST_SLICED]
REFERENCE_ORDER = (R.TRAV_WAVE, R.TRAV_LANE, R.TRAV_NESTED)


def check_counts(trav, st, ost):
    assert (st.rays, st.tri_hits) == (ost.rays, ost.tri_hits)
    if trav in REFERENCE_ORDER:
        assert (st.box_tests, st.tri_tests) == (ost.box_tests, ost.tri_tests)
    else:
        assert 0 < st.box_tests and 0 < st.tri_tests
        if trav == R.TRAV_FAST_REPLAY:
            assert st.replays >= st.tri_hits  # every ray with a candidate went through the replay


@pytest.fixture(scope="module")
def tracer():
    t = R.Tracer(0)
    yield t
    t.close()


@pytest.fixture(scope="module")
def dragon2426(tracer):
    return product_sd_scene([WL.DRAGON_2426]), oracle_sd_scene([WL.DRAGON_2426])


@pytest.mark.parametrize("trav", TRAVS)
def test_primary_ids_dragon2426_400x225(tracer, dragon2426, trav):
    ps, os_ = dragon2426
    tracer.upload(ps)
    tracer.set_camera(WL.DRAGON_2426_CAMERA)
    tracer.set_options(bounce_limit=8, jitter=0.0, collect_stats=True, traversal=trav)
    img, prim, st = tracer.trace(400, 225, 1, 1, want_primary=True)
    opt = O.make_options(bounce_limit=8, jitter=0.0, trig=O.TRIG_POLY)
    oimg, oprim, ost = os_.render(oracle_camera(WL.DRAGON_2426_CAMERA), opt, 400, 225, 1, 1, want_primary=True)
    assert (prim != R.MISS).sum() > 1000
    assert np.count_nonzero(prim != oprim) == 0
    check_counts(trav, st, ost)
    assert np.array_equal(bits(img), bits(oimg))


def test_bvh_acceleration_off(tracer, dragon2426):
    """RT_TRAV_BRUTE = every triangle, no box (the reference's "BVH Acceleration" debugging switch, BVHUI.hpp:14, given to
    the flat scene as well): same hits as the oracle's brute-force scan, and -- the reference's own check for its BVH,
    closest-hit t must agree -- the same image as the tree walk except where two triangles tie."""
    ps, os_ = dragon2426
    tracer.upload(ps)
    tracer.set_camera(WL.DRAGON_2426_CAMERA)
    tracer.set_options(bounce_limit=3, seed=7, collect_stats=True, traversal=R.TRAV_BRUTE)
    img, prim, st = tracer.trace(160, 90, 1, 2, want_primary=True)
    opt = O.make_options(bounce_limit=3, seed=7, trig=O.TRIG_POLY, traversal=O.TRAV_BRUTE)
    oimg, oprim, ost = os_.render(oracle_camera(WL.DRAGON_2426_CAMERA), opt, 160, 90, 1, 2, want_primary=True)
    assert (prim != R.MISS).sum() > 200
    assert np.count_nonzero(prim != oprim) == 0
    assert (st.rays, st.box_tests, st.tri_tests, st.tri_hits) == (ost.rays, 0, ost.tri_tests, ost.tri_hits)
    assert st.tri_tests == st.rays * 2426
    assert np.array_equal(bits(img), bits(oimg))
    tracer.set_options(bounce_limit=3, seed=7, traversal=R.TRAV_FAST)
    timg, tprim, _ = tracer.trace(160, 90, 1, 2, want_primary=True)
    assert np.count_nonzero(tprim != prim) <= 2 and np.mean(bits(timg) != bits(img)) < 1e-3


def test_primary_ids_dragon2426_1080p(tracer, dragon2426):
    ps, os_ = dragon2426
    tracer.upload(ps)
    for cam in (WL.DRAGON_2426_CAMERA, WL.DEFAULT_CAMERA):
        tracer.set_camera(cam)
        tracer.set_options(bounce_limit=1, jitter=0.0)
        _, prim, st = tracer.trace(1920, 1080, 1, 1, want_image=False, want_primary=True)
        opt = O.make_options(bounce_limit=1, jitter=0.0)
        _, oprim, ost = os_.render(oracle_camera(cam), opt, 1920, 1080, 1, 1, want_primary=True)
        assert st.rays == ost.rays == 1920 * 1080
        assert np.count_nonzero(prim != oprim) == 0


@pytest.mark.parametrize("trav", TRAVS)
def test_shaded_image_bit_exact(tracer, dragon2426, trav):
    ps, os_ = dragon2426
    tracer.upload(ps)
    tracer.set_camera(WL.DRAGON_2426_CAMERA)
    tracer.set_options(bounce_limit=8, seed=7, jitter=0.001, collect_stats=True, traversal=trav)
    img, _, st = tracer.trace(320, 180, 1, 8)
    opt = O.make_options(bounce_limit=8, seed=7, jitter=0.001, trig=O.TRIG_POLY)
    oimg, _, ost = os_.render(oracle_camera(WL.DRAGON_2426_CAMERA), opt, 320, 180, 1, 8)
    check_counts(trav, st, ost)
    assert np.array_equal(bits(img), bits(oimg))
    # against libm cos/sin: the stated per-pixel tolerance
    opt.trig = O.TRIG_LIBM
    limg, _, _ = os_.render(oracle_camera(WL.DRAGON_2426_CAMERA), opt, 320, 180, 1, 8)
    ok = np.all(np.abs(img - limg) <= 1e-4 + 1e-3 * np.abs(limg), axis=-1)
    assert ok.mean() >= 0.999
    assert psnr(img, limg) >= 45.0


def test_progressive_equals_batched(tracer, dragon2426):
    """sampleCount protocol: 1 spp per call for n calls == n spp in one call (same running mean)."""
    ps, _ = dragon2426
    tracer.upload(ps)
    tracer.set_camera(WL.DRAGON_2426_CAMERA)
    tracer.set_options(bounce_limit=4, seed=3)
    whole, _, _ = tracer.trace(256, 144, 1, 6)
    for n in range(1, 7):
        part, _, _ = tracer.trace(256, 144, n, 1)
    assert np.array_equal(bits(whole), bits(part))
    a, _, _ = tracer.trace(256, 144, 1, 2)
    b, _, _ = tracer.trace(256, 144, 3, 4)
    assert np.array_equal(bits(whole), bits(b))


@pytest.mark.parametrize("trav", TRAVS)
def test_tile_split_union(tracer, dragon2426, trav):
    ps, _ = dragon2426
    tracer.upload(ps)
    tracer.set_camera(WL.DRAGON_2426_CAMERA)
    tracer.set_options(bounce_limit=4, seed=5, traversal=trav)
    whole, _, st = tracer.trace(250, 141, 1, 2)  # not a multiple of the 8x8 tile
    acc = np.zeros_like(whole)
    rays = 0
    for r in range(3):
        tracer.set_options(bounce_limit=4, seed=5, tile_rank=r, tile_world=3, traversal=trav)
        part, _, s = tracer.trace(250, 141, 1, 2)
        assert np.all((acc == 0) | (part == 0))
        acc += part
        rays += s.rays
    assert rays == st.rays
    assert np.array_equal(bits(whole), bits(acc))


@pytest.mark.parametrize("trav", TRAVS)
def test_dragon18520_small(tracer, trav):
    ps, os_ = product_sd_scene([WL.DRAGON_18520]), oracle_sd_scene([WL.DRAGON_18520])
    tracer.upload(ps)
    tracer.set_camera(WL.DRAGON_18520_CAMERA)
    tracer.set_options(bounce_limit=8, seed=1, collect_stats=True, traversal=trav)
    img, prim, st = tracer.trace(240, 135, 1, 2, want_primary=True)
    opt = O.make_options(bounce_limit=8, seed=1, trig=O.TRIG_POLY)
    oimg, oprim, ost = os_.render(oracle_camera(WL.DRAGON_18520_CAMERA), opt, 240, 135, 1, 2, want_primary=True)
    assert np.count_nonzero(prim != oprim) == 0
    check_counts(trav, st, ost)
    assert np.array_equal(bits(img), bits(oimg))


@pytest.mark.parametrize("trav", [R.TRAV_FAST, R.TRAV_FAST_REPLAY, R.TRAV_FAST_REFILL, R.TRAV_FAST_SLICED, R.TRAV_LANE])
def test_dragon_field_k2(tracer, trav):
    """A 2 x 2 grid of translated dragon copies flattened into one triangle list (the small case of the
    instanced dragon field), looked at along the rows so that rays cross several copies."""
    tris, _ = objparse.load_obj(WL.DRAGON_2426)
    grid = WL.dragon_grid_tris(tris, 2)
    ps = R.SdScene().add_model(grid).finalize()
    os_ = O.SdScene()
    os_.add_model(grid, [len(grid)])
    os_.finalize()
    tracer.upload(ps)
    for cam in (R.Camera(1.0, (0.12, 0.5, 0.9), 2.0, 16.0 / 9.0, (0.12, 0.07, 0.11)),
                R.Camera(1.0, (-0.3, 0.08, 0.0), 2.0, 16.0 / 9.0, (0.3, 0.06, 0.0))):
        tracer.set_camera(cam)
        tracer.set_options(bounce_limit=8, seed=11, collect_stats=True, traversal=trav)
        img, prim, st = tracer.trace(256, 144, 1, 2, want_primary=True)
        opt = O.make_options(bounce_limit=8, seed=11, trig=O.TRIG_POLY)
        oimg, oprim, ost = os_.render(oracle_camera(cam), opt, 256, 144, 1, 2, want_primary=True)
        assert (prim != R.MISS).sum() > 500
        assert np.count_nonzero(prim != oprim) == 0
        check_counts(trav, st, ost)
        assert np.array_equal(bits(img), bits(oimg))


def test_scene_beyond_the_l2s_every_fast_variant_agrees(tracer):
    """5 x 5 copies of the 18 520-triangle dragon (463 000 triangles: fast tree + positions = 52 MB, more than the L2s hold),
    the size from which RT_TRAV_FAST runs the re-fill kernel; the oracle is too slow for this scene at a useful image
    size, so the fast variants, which are separate kernels, are held to each other bit for bit."""
    tris, _ = objparse.load_obj(WL.DRAGON_18520)
    ps = R.SdScene().add_model(WL.dragon_grid_tris(tris, 5)).finalize()
    tracer.upload(ps)
    tracer.set_camera(WL.field_camera(5))
    out = {}
    for trav in (R.TRAV_FAST, R.TRAV_FAST_REFILL, R.TRAV_FAST_SLICED, R.TRAV_FAST_QUEUED, R.TRAV_FAST_INLINE):
        tracer.set_options(bounce_limit=8, seed=5, collect_stats=True, traversal=trav)
        out[trav] = tracer.trace(960, 540, 1, 2, want_primary=True)
    fi, fp, fs = out[R.TRAV_FAST]
    assert (fp != R.MISS).sum() > 50000 and fs.replays > 0
    for trav, (oi, op, os2) in out.items():
        assert (os2.rays, os2.box_tests, os2.tri_tests, os2.tri_hits) == (fs.rays, fs.box_tests, fs.tri_tests, fs.tri_hits), trav
        assert np.array_equal(op, fp), trav
        assert np.array_equal(bits(oi), bits(fi)), trav


def test_full_size_properties_fast_vs_reference_order(tracer):
    """At the benchmark's full size the oracle is too slow to be the checker; the two device walks are
    independent implementations of the same contract, so their images must agree bit for bit."""
    ps = product_sd_scene([WL.DRAGON_18520])
    tracer.upload(ps)
    tracer.set_camera(WL.DRAGON_18520_CAMERA)
    out = {}
    for trav in (R.TRAV_FAST, R.TRAV_FAST_INLINE, R.TRAV_FAST_REFILL, R.TRAV_FAST_SLICED, R.TRAV_WAVE):
        tracer.set_options(bounce_limit=8, seed=1, traversal=trav)
        out[trav] = tracer.trace(1920, 1080, 1, 2, want_primary=True)
    fi, fp, fs = out[R.TRAV_FAST]
    for trav in (R.TRAV_FAST_INLINE, R.TRAV_FAST_REFILL, R.TRAV_FAST_SLICED, R.TRAV_WAVE):
        wi, wp, ws = out[trav]
        assert fs.rays == ws.rays
        assert np.array_equal(fp, wp)
        assert np.array_equal(bits(fi), bits(wi))
    # alpha = fraction of samples that escaped to the sky, in [0, 1]; colours are finite and non-negative
    assert np.isfinite(fi).all() and fi.min() >= 0.0 and fi[..., 3].max() <= 1.0


def test_hand_made_tree_with_boxes_that_do_not_nest(tracer):
    """The fast walk is exact only when interior boxes contain their children's.  A tree that breaks this (here: the
    root's box shrunk to half the model) is walked in reference order by RT_TRAV_AUTO and refused by RT_TRAV_FAST."""
    sc = product_sd_scene([WL.DRAGON_2426])
    nodes, tris = sc.nodes().copy(), sc.tris()
    tracer.upload_arrays(nodes, sc.root, tris, np.zeros(0), np.zeros(0), sc.order)
    tracer.set_camera(WL.DRAGON_2426_CAMERA)
    tracer.set_options(bounce_limit=2, seed=4)
    good, gprim, _ = tracer.trace(192, 108, 1, 2, want_primary=True)
    nodes["pmax"][sc.root][0] = 0.0   # the root now excludes everything at x > 0
    tracer.upload_arrays(nodes, sc.root, tris, np.zeros(0), np.zeros(0), sc.order)
    tracer.set_options(bounce_limit=2, seed=4)                              # AUTO
    auto, aprim, _ = tracer.trace(192, 108, 1, 2, want_primary=True)
    tracer.set_options(bounce_limit=2, seed=4, traversal=R.TRAV_LANE)
    lane, lprim, _ = tracer.trace(192, 108, 1, 2, want_primary=True)
    assert np.array_equal(aprim, lprim) and np.array_equal(bits(auto), bits(lane))
    assert np.count_nonzero(aprim != gprim) > 100                           # the shrunk box does change the result
    tracer.set_options(bounce_limit=2, seed=4, traversal=R.TRAV_FAST)
    with pytest.raises(R.TracerError) as e:
        tracer.trace(192, 108, 1, 2)
    assert e.value.code == -5


def test_errors(tracer):
    t2 = R.Tracer(0)
    t2.set_camera(WL.DEFAULT_CAMERA)
    with pytest.raises(R.TracerError) as e:
        t2.trace(64, 36)
    assert e.value.code == -4 and "Scene is not loaded." in str(e.value)
    with pytest.raises(R.TracerError):
        t2.set_options(tile_rank=2, tile_world=2)
    nodes = product_sd_scene([WL.DRAGON_2426]).nodes().copy()
    nodes["left"][-1] = 10 ** 6
    with pytest.raises(R.TracerError) as e:
        t2.upload_arrays(nodes, len(nodes) - 1, product_sd_scene([WL.DRAGON_2426]).tris(), np.zeros(0), np.zeros(0), 0)
    assert e.value.code == -5
    t2.close()


def test_counting_the_walk_that_is_timed(tracer):
    """collect_stats = 2 counts the walk a trace without statistics makes (wide records, distance limit): the same image,
    rays and hits as the exact walk's counting variant, fewer triangle tests (leaves beyond the limit are skipped), and box
    tests counted per child box."""
    ps = product_sd_scene([WL.DRAGON_18520])
    tracer.upload(ps)
    tracer.set_camera(WL.DRAGON_18520_CAMERA)
    out = {}
    for mode in (0, 1, 2):
        tracer.set_options(bounce_limit=6, seed=3, collect_stats=mode)
        img, _, st = tracer.trace(320, 180, 1, 4)
        out[mode] = (img.copy(), st)
    assert np.array_equal(bits(out[0][0]), bits(out[1][0])) and np.array_equal(bits(out[0][0]), bits(out[2][0]))
    assert out[1][1].rays == out[2][1].rays == out[0][1].rays and out[1][1].tri_hits == out[2][1].tri_hits
    assert 0 < out[2][1].tri_tests < out[1][1].tri_tests
    assert out[2][1].box_tests > 0 and out[0][1].box_tests == 0
