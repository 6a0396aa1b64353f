"""The camera stage (RT_CAMERA_SEPARATE: camera rays and the walk's first slab test in a kernel of their own, sky samples
folded without ever being queued) against the fused first launch (RT_CAMERA_FUSED) and the oracle: same bits -- images,
primary ids, ray and test counts -- for every fast walk, on flat and pointer scenes, whole frames and tile shares, one call
and the progressive protocol, ragged frames, bounce limits 0 and 1."""
import numpy as np
import pytest

from helpers import O, R, WL, bits, objparse, oracle_camera, oracle_old_scene, oracle_sd_scene, product_old_scene, product_sd_scene

pytestmark = pytest.mark.gpu

FAST = [R.TRAV_FAST, R.TRAV_FAST_REPLAY, R.TRAV_FAST_REFILL, R.TRAV_FAST_QUEUED, R.TRAV_FAST_INLINE, R.TRAV_FAST_SLICED]
# the dragon fills the frame of its own camera; from further away most camera rays miss its root box
FAR_CAMERA = R.Camera(1.0, (0.0, 0.10, 0.9), 2.0, 16.0 / 9.0, (-0.006, 0.073, -0.014))


@pytest.fixture(scope="module")
def tracer():
    t = R.Tracer(0)
    yield t
    t.close()


@pytest.fixture(scope="module")
def dragon():
    return product_sd_scene([WL.DRAGON_2426]), oracle_sd_scene([WL.DRAGON_2426])


@pytest.mark.parametrize("trav", FAST)
def test_flat_scene_both_stages_equal_the_oracle(tracer, dragon, trav):
    ps, os_ = dragon
    tracer.upload(ps)
    for cam, (w, h) in ((FAR_CAMERA, (331, 187)), (WL.DRAGON_2426_CAMERA, (200, 112))):
        tracer.set_camera(cam)
        opt = O.make_options(bounce_limit=8, seed=7, trig=O.TRIG_POLY)
        oimg, oprim, ost = os_.render(oracle_camera(cam), opt, w, h, 1, 5, want_primary=True)
        res = {}
        for stage in (R.CAMERA_SEPARATE, R.CAMERA_FUSED, R.CAMERA_AUTO):
            tracer.set_options(bounce_limit=8, seed=7, collect_stats=True, traversal=trav, camera_stage=stage)
            img, prim, st = tracer.trace(w, h, 1, 5, want_primary=True)
            assert np.array_equal(prim, oprim), stage
            assert np.array_equal(bits(img), bits(oimg)), stage
            assert (st.rays, st.tri_hits) == (ost.rays, ost.tri_hits), stage
            res[stage] = st
        a, b = res[R.CAMERA_SEPARATE], res[R.CAMERA_FUSED]
        assert (a.box_tests, a.tri_tests, a.replays) == (b.box_tests, b.tri_tests, b.replays)


def test_tile_shares_and_progressive_calls(tracer, dragon):
    ps, _ = dragon
    tracer.upload(ps)
    tracer.set_camera(FAR_CAMERA)
    tracer.set_options(bounce_limit=4, seed=5, camera_stage=R.CAMERA_FUSED)
    whole, wprim, st = tracer.trace(250, 141, 1, 70, want_primary=True)   # 70 samples: two chunks (64 + 6)
    acc, rays = np.zeros_like(whole), 0
    for r in range(3):
        tracer.set_options(bounce_limit=4, seed=5, tile_rank=r, tile_world=3, camera_stage=R.CAMERA_SEPARATE)
        part, _, s = tracer.trace(250, 141, 1, 70)
        assert np.all((acc == 0) | (part == 0))
        acc += part
        rays += s.rays
    assert rays == st.rays and np.array_equal(bits(whole), bits(acc))
    tracer.set_options(bounce_limit=4, seed=5, camera_stage=R.CAMERA_SEPARATE)
    for first, n in ((1, 1), (2, 3), (5, 64), (69, 2)):
        part, pprim, _ = tracer.trace(250, 141, first, n, want_primary=(first == 1))
        if first == 1:
            assert np.array_equal(pprim, wprim)
    assert np.array_equal(bits(whole), bits(part))


@pytest.mark.parametrize("bounce", [0, 1, 2])
def test_bounce_limits(tracer, dragon, bounce):
    ps, os_ = dragon
    tracer.upload(ps)
    tracer.set_camera(FAR_CAMERA)
    oimg, oprim, ost = os_.render(oracle_camera(FAR_CAMERA), O.make_options(bounce_limit=bounce, seed=2, trig=O.TRIG_POLY), 96, 54, 1, 3,
                                  want_primary=True)
    for stage in (R.CAMERA_SEPARATE, R.CAMERA_FUSED):
        tracer.set_options(bounce_limit=bounce, seed=2, collect_stats=True, camera_stage=stage)
        img, prim, st = tracer.trace(96, 54, 1, 3, want_primary=True)
        assert np.array_equal(prim, oprim) and np.array_equal(bits(img), bits(oimg)) and st.rays == ost.rays


@pytest.mark.parametrize("trav", [R.TRAV_FAST, R.TRAV_FAST_REFILL, R.TRAV_FAST_INLINE, R.TRAV_FAST_SLICED])
def test_pointer_scene(tracer, trav):
    tris, _ = objparse.load_obj(WL.CYLINDER)
    spheres = WL.small_sphere_scene()[1:]          # without the ground sphere the scene covers part of the frame only
    meshes = [(tris, R.MAT_LAMBERTIAN, (0.8, 0.3, 0.3, 1.0), 0.0)]
    ps, os_ = product_old_scene(spheres, meshes), oracle_old_scene(spheres, meshes)
    cam = R.Camera(3.19, (13.0, 2.0, 3.0), 2.0, 16.0 / 9.0, (0.0, 0.5, 0.0))
    tracer.upload(ps)
    tracer.set_camera(cam)
    opt = O.make_options(bounce_limit=8, seed=3, trig=O.TRIG_POLY, accum=O.ACCUM_FORWARD)
    oimg, oprim, ost = os_.render(oracle_camera(cam), opt, 200, 113, 1, 4, want_primary=True)
    assert 0.02 < (oprim != R.MISS).mean() < 0.9
    for stage in (R.CAMERA_SEPARATE, R.CAMERA_FUSED):
        tracer.set_options(bounce_limit=8, estimator=R.EST_OLD_RECURSIVE, seed=3, collect_stats=True, traversal=trav,
                           camera_stage=stage)
        img, prim, st = tracer.trace(200, 113, 1, 4, want_primary=True)
        assert np.array_equal(prim, oprim), stage
        assert np.array_equal(bits(img), bits(oimg)), stage
        assert (st.rays, st.sph_tests > 0) == (ost.rays, True)


def test_chunk_of_more_than_64_samples(tracer, dragon):
    """A small frame (or a share of a frame) is traced in chunks of up to 512 samples: the camera stage then keeps several
    64-bit words of its sky mask per pixel.  200 samples in one call against the oracle, and against calls of 64."""
    ps, os_ = dragon
    tracer.upload(ps)
    tracer.set_camera(FAR_CAMERA)
    oimg, oprim, ost = os_.render(oracle_camera(FAR_CAMERA), O.make_options(bounce_limit=4, seed=21, trig=O.TRIG_POLY), 96, 54, 1, 200,
                                  want_primary=True)
    for stage in (R.CAMERA_SEPARATE, R.CAMERA_FUSED):
        tracer.set_options(bounce_limit=4, seed=21, camera_stage=stage)
        img, prim, st = tracer.trace(96, 54, 1, 200, want_primary=True)
        assert np.array_equal(prim, oprim) and np.array_equal(bits(img), bits(oimg)) and st.rays == ost.rays, stage
        for first in (1, 65, 129, 193):
            part, _, _ = tracer.trace(96, 54, first, min(64, 201 - first))
        assert np.array_equal(bits(part), bits(oimg)), stage
    # a third of the tiles of a larger frame, 300 samples: the share's chunk holds them all
    tracer.set_options(bounce_limit=3, seed=4)
    whole, _, wst = tracer.trace(331, 187, 1, 300)
    acc, rays = np.zeros_like(whole), 0
    for r in range(3):
        tracer.set_options(bounce_limit=3, seed=4, tile_rank=r, tile_world=3, camera_stage=R.CAMERA_SEPARATE)
        part, _, s = tracer.trace(331, 187, 1, 300)
        assert np.all((acc == 0) | (part == 0))
        acc += part
        rays += s.rays
    assert rays == wst.rays and np.array_equal(bits(whole), bits(acc))


@pytest.mark.parametrize("size", [(16, 8), (56, 8), (24, 16)])
def test_few_tiles_many_samples_per_chunk(tracer, dragon, size):
    """A share of two to seven tiles with 512 samples in ONE chunk, every camera ray past the root box: each (tile, 64
    samples) unit of the camera stage queues up to 4096 records at once, so the units must be spread over the sub-queues
    (one tile's eight groups landing in one sub-queue would overrun it into its neighbour's records).  Both estimators,
    SEPARATE against FUSED against the oracle, 512 and 500 samples (a partial last group)."""
    w, h = size
    ps, os_ = dragon
    cam = R.Camera(1.0, (0.0, 0.07, 0.02), 2.0, 16.0 / 9.0, (-0.006, 0.073, -0.014))   # inside the dragon's box: every camera ray gets past the root record
    tracer.upload(ps)
    tracer.set_camera(cam)
    for spp in (512, 500):
        oimg, oprim, ost = os_.render(oracle_camera(cam), O.make_options(bounce_limit=5, seed=31, trig=O.TRIG_POLY, threads=8), w, h, 1, spp,
                                      want_primary=True)
        for trav in (R.TRAV_FAST, R.TRAV_FAST_QUEUED, R.TRAV_FAST_REFILL):
            for stage in (R.CAMERA_SEPARATE, R.CAMERA_FUSED):
                tracer.set_options(bounce_limit=5, seed=31, traversal=trav, camera_stage=stage)
                img, prim, st = tracer.trace(w, h, 1, spp, want_primary=True)
                assert st.rays == ost.rays, (spp, trav, stage)
                assert np.array_equal(prim, oprim) and np.array_equal(bits(img), bits(oimg)), (spp, trav, stage)
    # the recursive estimator (four-float4 records): spheres all round the camera
    spheres = WL.small_sphere_scene()
    pso, oso = product_old_scene(spheres), oracle_old_scene(spheres)
    cam2 = R.Camera(1.0, (0.0, 1.0, 4.0), 2.0, 16.0 / 9.0, (0.0, 1.0, 0.0))
    tracer.upload(pso)
    tracer.set_camera(cam2)
    opt = O.make_options(bounce_limit=5, seed=32, trig=O.TRIG_POLY, accum=O.ACCUM_FORWARD, threads=8)
    oimg, oprim, ost = oso.render(oracle_camera(cam2), opt, w, h, 1, 512, want_primary=True)
    for trav in (R.TRAV_FAST, R.TRAV_FAST_REFILL):
        for stage in (R.CAMERA_SEPARATE, R.CAMERA_FUSED):
            tracer.set_options(bounce_limit=5, estimator=R.EST_OLD_RECURSIVE, seed=32, traversal=trav, camera_stage=stage)
            img, prim, st = tracer.trace(w, h, 1, 512, want_primary=True)
            assert st.rays == ost.rays, (trav, stage)
            assert np.array_equal(prim, oprim) and np.array_equal(bits(img), bits(oimg)), (trav, stage)


@pytest.mark.parametrize("jitter", [0.001, 0.02, 0.2])
def test_tiles_that_cannot_see_the_scene_are_skipped_safely(tracer, dragon, jitter):
    """The camera stage leaves out the tiles outside the widened projection of the scene's box: with the scene in a corner
    of the frame, partly outside it, straight ahead and seen from far away -- and with jitter up to two hundred times the
    reference's, which widens the projection (and at 0.2 switches the shortcut off) -- the frames are those of the fused
    first launch, bit for bit, and the ray counts agree."""
    ps, _ = dragon
    tracer.upload(ps)
    cams = [R.Camera(1.0, (0.0, 0.10, 0.9), 2.0, 16.0 / 9.0, (0.45, 0.30, -0.014)),      # scene in the lower left corner
            R.Camera(1.0, (0.0, 0.10, 0.9), 2.0, 16.0 / 9.0, (-0.9, -0.35, -0.014)),     # partly outside the frame
            R.Camera(1.0, (0.0, 0.10, 3.0), 2.0, 16.0 / 9.0, (-0.006, 0.073, -0.014)),   # small, in the middle
            R.Camera(0.4, (0.3, 0.25, 0.6), 2.0, 16.0 / 9.0, (-0.006, 0.073, -0.014))]   # wide angle, from the side
    for cam in cams:
        tracer.set_camera(cam)
        out = {}
        for stage in (R.CAMERA_SEPARATE, R.CAMERA_FUSED):
            tracer.set_options(bounce_limit=3, seed=11, jitter=jitter, camera_stage=stage)
            out[stage] = tracer.trace(640, 360, 1, 3, want_primary=True)
        (ia, pa, sa), (ib, pb, sb) = out[R.CAMERA_SEPARATE], out[R.CAMERA_FUSED]
        assert np.array_equal(pa, pb) and np.array_equal(bits(ia), bits(ib))
        assert sa.rays == sb.rays
        assert (pa != R.MISS).sum() > 50          # the scene is in the frame
    # ... and against the oracle itself, the mask at work (no statistics asked for), the reference's and a larger jitter
    if jitter <= 0.02:
        _, os_ = dragon
        cam = cams[2]
        tracer.set_camera(cam)
        tracer.set_options(bounce_limit=3, seed=11, jitter=jitter, camera_stage=R.CAMERA_SEPARATE)
        img, prim, st = tracer.trace(320, 180, 1, 2, want_primary=True)
        opt = O.make_options(bounce_limit=3, seed=11, jitter=jitter, trig=O.TRIG_POLY)
        oimg, oprim, ost = os_.render(oracle_camera(cam), opt, 320, 180, 1, 2, want_primary=True)
        assert np.array_equal(prim, oprim) and np.array_equal(bits(img), bits(oimg)) and st.rays == ost.rays
