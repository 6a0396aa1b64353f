"""GPU parity tests for the pointer-based ("old") scene path: spheres, meshes with an inner BVH,
Lambertian / metal / emitter / dielectric materials, Russian roulette -- through the C ABI.

Bar: bit-exact against the oracle evaluated with the same front-to-back product order
(ORA_ACCUM_FORWARD); the oracle's recursive form (the reference's own order of multiplications) must
agree with that to 1e-6 relative.
"""
import numpy as np
import pytest

from helpers import O, R, WL, bits, objparse, oracle_camera, oracle_old_scene, product_old_scene, psnr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tracer():
    t = R.Tracer(0)
    yield t
    t.close()


REPLAYS = {}


def _compare(tracer, ps, os_, cam, w, h, spp, bounce, seed, jitter=0.001):
    tracer.upload(ps)
    tracer.set_camera(cam)
    opt = O.make_options(bounce_limit=bounce, seed=seed, jitter=jitter, trig=O.TRIG_POLY, accum=O.ACCUM_FORWARD)
    oimg, oprim, ost = os_.render(oracle_camera(cam), opt, w, h, 1, spp, want_primary=True)
    for trav in (R.TRAV_WAVE, R.TRAV_LANE, R.TRAV_NESTED, R.TRAV_FAST, R.TRAV_FAST_REPLAY, R.TRAV_FAST_REFILL, R.TRAV_FAST_QUEUED, R.TRAV_FAST_INLINE, R.TRAV_FAST_SLICED):   # every traversal variant: same bits
        tracer.set_options(bounce_limit=bounce, estimator=R.EST_OLD_RECURSIVE, seed=seed, jitter=jitter,
                           collect_stats=True, traversal=trav)
        img, prim, st = tracer.trace(w, h, 1, spp, want_primary=True)
        assert np.count_nonzero(prim != oprim) == 0, trav
        assert st.rays == ost.rays, trav
        if trav in (R.TRAV_WAVE, R.TRAV_LANE, R.TRAV_NESTED):  # these perform the reference's own tests
            assert (st.box_tests, st.tri_tests, st.sph_tests) == (ost.box_tests, ost.tri_tests, ost.sph_tests), trav
        assert np.array_equal(bits(img), bits(oimg)), trav
        REPLAYS[trav] = REPLAYS.get(trav, 0) + st.replays
    # "BVH Acceleration" unticked (BVHUI.hpp:14): Scene::intersectClosest + Mesh::intersect without the inner BVH
    opt.traversal = O.TRAV_BRUTE
    bimg, bprim, bst = os_.render(oracle_camera(cam), opt, w, h, 1, spp, want_primary=True)
    opt.traversal = O.TRAV_LOOP
    tracer.set_options(bounce_limit=bounce, estimator=R.EST_OLD_RECURSIVE, seed=seed, jitter=jitter, collect_stats=True,
                       traversal=R.TRAV_BRUTE)
    gimg, gprim, gst = tracer.trace(w, h, 1, spp, want_primary=True)
    assert np.count_nonzero(gprim != bprim) == 0
    assert (gst.rays, gst.box_tests, gst.tri_tests, gst.sph_tests) == (bst.rays, 0, bst.tri_tests, bst.sph_tests)
    assert np.array_equal(bits(gimg), bits(bimg))
    return img, opt


def test_small_spheres_all_materials(tracer):
    sp = WL.small_sphere_scene()
    ps, os_ = product_old_scene(sp), oracle_old_scene(sp)
    cam = R.Camera(3.19, (13.0, 2.0, 3.0), 2.0, 16.0 / 9.0, (0.0, 0.5, 0.0))
    img, opt = _compare(tracer, ps, os_, cam, 400, 225, 8, 8, seed=1)
    assert img[..., :3].max() > 1.0  # the emitter is visible
    # the reference's recursive product order differs from the loop's only by rounding
    opt.accum = O.ACCUM_RECURSIVE
    rimg, _, _ = os_.render(oracle_camera(cam), opt, 400, 225, 1, 8)
    assert np.allclose(img, rimg, rtol=1e-6, atol=1e-7)


def test_rtiow_scene(tracer):
    sp = WL.rtiow_spheres()
    ps, os_ = product_old_scene(sp), oracle_old_scene(sp)
    img, opt = _compare(tracer, ps, os_, WL.RTIOW_CAMERA, 480, 270, 4, 8, seed=42)
    opt.trig = O.TRIG_LIBM
    opt.accum = O.ACCUM_RECURSIVE
    limg, _, _ = os_.render(oracle_camera(WL.RTIOW_CAMERA), opt, 480, 270, 1, 4)
    ok = np.all(np.abs(img - limg) <= 1e-4 + 1e-3 * np.abs(limg), axis=-1)
    assert ok.mean() >= 0.999


def test_mesh_with_inner_bvh_and_spheres(tracer):
    """Config 1: the bundled 124-triangle cylinder (SphereSingle.obj) as a Mesh object next to spheres."""
    tris, _ = objparse.load_obj(WL.CYLINDER)
    meshes = [(tris, R.MAT_LAMBERTIAN, (0.8, 0.3, 0.3, 1.0), 0.0)]
    sp = WL.small_sphere_scene()
    ps, os_ = product_old_scene(sp, meshes), oracle_old_scene(sp, meshes)
    cam = R.Camera(3.19, (13.0, 2.0, 3.0), 2.0, 16.0 / 9.0, (0.0, 0.5, 0.0))
    _compare(tracer, ps, os_, cam, 400, 225, 8, 8, seed=1)
    cam2 = R.Camera(1.0, (0.5, 1.5, 6.0), 2.0, 16.0 / 9.0, (0.0, 0.8, 0.0))
    _compare(tracer, ps, os_, cam2, 200, 113, 3, 4, seed=9)


def test_two_meshes_metal_and_glass(tracer):
    tris, _ = objparse.load_obj(WL.DRAGON_2426)
    t2 = tris.copy().reshape(-1, 3, 8)
    t2[:, :, 0] += 0.25
    meshes = [(tris, R.MAT_METAL, (0.8, 0.8, 0.9, 1.0), 0.1), (t2.reshape(-1, 24), R.MAT_DIELECTRIC, (1, 1, 1, 1), 1.5)]
    sp = [((0.0, -100.0, 0.0), 100.0, R.MAT_LAMBERTIAN, (0.5, 0.5, 0.5, 1.0), 0.0)]
    ps, os_ = product_old_scene(sp, meshes), oracle_old_scene(sp, meshes)
    cam = R.Camera(1.0, (0.12, 0.12, 0.45), 2.0, 16.0 / 9.0, (0.12, 0.07, 0.0))
    _compare(tracer, ps, os_, cam, 320, 180, 4, 8, seed=5)


def test_ties_and_rays_starting_inside(tracer):
    """Duplicated spheres and a duplicated mesh (equal-distance ties: the first in the reference's visiting order
    wins), concentric spheres, and a camera inside a large Lambertian sphere and inside a glass one (a ray that
    starts inside never hits a non-dielectric sphere: near root only, Objects.hpp:57-60)."""
    tris, _ = objparse.load_obj(WL.CYLINDER)
    sp = WL.small_sphere_scene()
    sp = sp + [sp[2], sp[3], sp[2]]                                       # exact duplicates
    sp.append(((0.0, 1.0, 0.0), 0.5, R.MAT_METAL, (0.9, 0.9, 0.9, 1.0), 0.2))   # inside the glass sphere
    sp.append(((13.0, 2.0, 3.0), 2.5, R.MAT_LAMBERTIAN, (0.3, 0.6, 0.3, 1.0), 0.0))  # around the first camera
    meshes = [(tris, R.MAT_LAMBERTIAN, (0.8, 0.3, 0.3, 1.0), 0.0), (tris.copy(), R.MAT_METAL, (0.7, 0.7, 0.9, 1.0), 0.3)]
    ps, os_ = product_old_scene(sp, meshes), oracle_old_scene(sp, meshes)
    _compare(tracer, ps, os_, R.Camera(3.19, (13.0, 2.0, 3.0), 2.0, 16.0 / 9.0, (0.0, 0.5, 0.0)), 200, 113, 3, 8, seed=2)
    _compare(tracer, ps, os_, R.Camera(1.0, (0.0, 1.0, 0.7), 2.0, 16.0 / 9.0, (4.0, 1.0, 0.0)), 200, 113, 3, 8, seed=4)   # inside the glass
    _compare(tracer, ps, os_, R.Camera(1.0, (0.5, 1.5, 6.0), 2.0, 16.0 / 9.0, (0.0, 0.8, 0.0)), 33, 17, 2, 1, seed=6)


def test_full_size_fast_vs_reference_order(tracer):
    """At the benchmark's full size (485 spheres, 1080p) the oracle is too slow to be the checker; the fast walk, its
    re-fill variant and the reference-order walk are independent implementations and must agree bit for bit.  The
    spheres touch their boxes, so this frame also sends thousands of rays through the ordered replay."""
    ps = product_old_scene(WL.rtiow_spheres())
    tracer.upload(ps)
    tracer.set_camera(WL.RTIOW_CAMERA)
    out = {}
    for trav in (R.TRAV_FAST, R.TRAV_FAST_REFILL, R.TRAV_FAST_QUEUED, R.TRAV_FAST_INLINE, R.TRAV_FAST_SLICED, R.TRAV_LANE):
        tracer.set_options(bounce_limit=8, estimator=R.EST_OLD_RECURSIVE, seed=42, traversal=trav)
        out[trav] = tracer.trace(1920, 1080, 1, 3, want_primary=True)
    fi, fp, fs = out[R.TRAV_FAST]
    assert fs.replays > 100
    for trav in (R.TRAV_FAST_REFILL, R.TRAV_FAST_QUEUED, R.TRAV_FAST_INLINE, R.TRAV_FAST_SLICED, R.TRAV_LANE):
        oi, op, os2 = out[trav]
        assert fs.rays == os2.rays
        assert np.array_equal(fp, op)
        assert np.array_equal(bits(fi), bits(oi))
    assert np.isfinite(fi).all() and fi.min() >= 0.0


def test_plugin_adaptors_drive_the_tracer(tracer):
    """RenderPipeline<SceneHIPContext, LoadSceneHIP, TraceSceneHIP> + headless TracerAsync: load once, one
    trace call per frame with sampleCount = 1, 2, 3 ... gives the n-sample running mean."""
    sp = WL.small_sphere_scene()
    ps, os_ = product_old_scene(sp), oracle_old_scene(sp)
    cam = R.Camera(3.19, (13.0, 2.0, 3.0), 2.0, 16.0 / 9.0, (0.0, 0.5, 0.0))
    ps.publish()
    h = R.PluginHarness(R.PluginHarness.MODE_OLD, 160, 90, cam, bounce_limit=8, seed=1, jitter=0.001)
    with pytest.raises(RuntimeError, match="Scene is not loaded."):
        h.render()
    h.reset()
    h.load()
    for _ in range(5):
        h.render()
    assert h.sample_count() == 6
    out = h.output()
    opt = O.make_options(bounce_limit=8, seed=1, jitter=0.001, trig=O.TRIG_POLY, accum=O.ACCUM_FORWARD)
    oimg, _, _ = os_.render(oracle_camera(cam), opt, 160, 90, 1, 5)
    assert np.array_equal(bits(out), bits(oimg))
    # RenderState::Dirty -> resetSamples: the mean restarts
    h.reset()
    h.render()
    oimg1, _, _ = os_.render(oracle_camera(cam), opt, 160, 90, 1, 1)
    assert np.array_equal(bits(h.output()), bits(oimg1))
    # the "BVH Acceleration" checkbox (BVHUI.hpp:14,25): unticked, the same frame comes from the brute-force scan
    try:
        h.set_bvh_accel(False)
        h.reset()
        h.render()
        opt.traversal = O.TRAV_BRUTE
        bimg, _, _ = os_.render(oracle_camera(cam), opt, 160, 90, 1, 1)
        assert np.array_equal(bits(h.output()), bits(bimg))
    finally:
        h.set_bvh_accel(True)
    # resize
    h.resize(96, 54)
    h.render()
    oimg2, _, _ = os_.render(oracle_camera(cam), opt, 96, 54, 1, 1)
    assert np.array_equal(bits(h.output()), bits(oimg2))
    h.close()


def test_plugin_sd_mode(tracer):
    from helpers import oracle_sd_scene, product_sd_scene
    ps, os_ = product_sd_scene([WL.DRAGON_2426]), oracle_sd_scene([WL.DRAGON_2426])
    ps.publish()
    h = R.PluginHarness(R.PluginHarness.MODE_SD, 192, 108, WL.DRAGON_2426_CAMERA, bounce_limit=4, seed=2, jitter=0.001)
    h.load()
    for _ in range(3):
        h.render()
    opt = O.make_options(bounce_limit=4, seed=2, jitter=0.001, trig=O.TRIG_POLY)
    oimg, _, _ = os_.render(oracle_camera(WL.DRAGON_2426_CAMERA), opt, 192, 108, 1, 3)
    assert np.array_equal(bits(h.output()), bits(oimg))
    h.close()
