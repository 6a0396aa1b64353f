"""The benchmark configurations at their stated sizes and scenes, on the GPU, through the C ABI.

Where the oracle finishes in seconds it is the checker (a small frame of the same scene and camera); at full size the
checkers are the independent device walks, which must agree bit for bit, and properties of the image."""
import numpy as np
import pytest

from helpers import O, R, WL, bits, objparse, oracle_camera, oracle_workload_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tracer():
    t = R.Tracer(0)
    yield t
    t.close()


@pytest.fixture(scope="module")
def config4():
    return WL.make_workload("dragon18520_4k")


def _image_properties(img):
    # alpha = fraction of samples that escaped to the sky, in [0, 1]; colours are finite and non-negative
    assert np.isfinite(img).all() and img.min() >= 0.0 and img[..., 3].max() <= 1.0


def test_config4_scene_small_frame_against_the_oracle(tracer, config4):
    """Configuration 4's scene (18 520-triangle dragon + the 888 960-triangle stand-in for the full dragon + ground quad,
    three models under one scene-level BVH) and camera, 64 x 36: ids, image and ray count equal the oracle's."""
    wl = config4
    assert len(wl.scene.tris()) == 18520 + 888960 + 2
    osc = oracle_workload_scene(wl)
    opt = O.make_options(bounce_limit=wl.bounce_limit, seed=wl.seed, jitter=wl.jitter, trig=O.TRIG_POLY, threads=16)
    oimg, oprim, ost = osc.render(oracle_camera(wl.camera), opt, 64, 36, 1, 2, want_primary=True)
    hit = oprim[oprim != R.MISS]
    # all three models are in the frame: the small dragon (ids < 18520), the stand-in, the ground (the last two ids)
    assert (hit < 18520).any() and ((hit >= 18520) & (hit < 18520 + 888960)).any() and (hit >= 18520 + 888960).any()
    tracer.upload(wl.scene)
    tracer.set_camera(wl.camera)
    for trav in (R.TRAV_FAST, R.TRAV_FAST_SLICED, R.TRAV_FAST_INLINE, R.TRAV_FAST_REPLAY, R.TRAV_LANE):
        tracer.set_options(bounce_limit=wl.bounce_limit, seed=wl.seed, jitter=wl.jitter, collect_stats=True, traversal=trav)
        img, prim, st = tracer.trace(64, 36, 1, 2, want_primary=True)
        assert np.array_equal(prim, oprim), trav
        assert (st.rays, st.tri_hits) == (ost.rays, ost.tri_hits), trav
        if trav == R.TRAV_LANE:
            assert (st.box_tests, st.tri_tests) == (ost.box_tests, ost.tri_tests)
        assert np.array_equal(bits(img), bits(oimg)), trav


def test_config4_at_4k_in_several_chunks(tracer, config4):
    """Configuration 4 at 3840 x 2160 with 50 samples per pixel: three chunks of at most 24 samples (a chunk holds at most
    192 Mi pixel samples), about 1.2 G rays.  The default walk (lane re-fill: the scene exceeds the L2s), the sliced walk
    and the plain walk are separate kernels and must agree bit for bit."""
    wl = config4
    tracer.upload(wl.scene)
    tracer.set_camera(wl.camera)
    out = {}
    for trav in (R.TRAV_FAST, R.TRAV_FAST_SLICED, R.TRAV_FAST_INLINE):
        tracer.set_options(bounce_limit=wl.bounce_limit, seed=wl.seed, jitter=wl.jitter, traversal=trav)
        img, prim, st = tracer.trace(wl.width, wl.height, 1, 50, want_primary=True)
        out[trav] = (bits(img).copy(), prim, st.rays)
    a = out[R.TRAV_FAST]
    assert a[2] > 1.5 * wl.width * wl.height * 50    # over the ground most paths bounce at least once
    for trav in (R.TRAV_FAST_SLICED, R.TRAV_FAST_INLINE):
        b = out[trav]
        assert a[2] == b[2] and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]), trav
    _image_properties(a[0].view(np.float32))
    assert (a[1] != R.MISS).mean() > 0.5


def test_bare_dragon_at_4k_in_several_chunks_fast_vs_reference_order(tracer):
    """The 18 520-triangle dragon alone at 3840 x 2160, 50 spp (three chunks, 450 M rays): the default walk (sliced), the
    plain fast walk and the wave walk, which performs the reference's own tests in the reference's order."""
    wl = WL.make_workload("dragon18520_bare_4k")
    tracer.upload(wl.scene)
    tracer.set_camera(wl.camera)
    out = {}
    for trav in (R.TRAV_FAST, R.TRAV_FAST_INLINE, R.TRAV_WAVE):
        tracer.set_options(bounce_limit=wl.bounce_limit, seed=wl.seed, jitter=wl.jitter, traversal=trav)
        img, prim, st = tracer.trace(wl.width, wl.height, 1, 50, want_primary=True)
        out[trav] = (bits(img).copy(), prim, st.rays)
    a = out[R.TRAV_FAST]
    for trav in (R.TRAV_FAST_INLINE, R.TRAV_WAVE):
        b = out[trav]
        assert a[2] == b[2] and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]), trav
    _image_properties(a[0].view(np.float32))


def test_million_triangle_field(tracer):
    """Configuration 5's small field: 8 x 8 dragons, 1 185 280 triangles in one flat list.  The default walk (lane re-fill)
    against the per-lane reference-order walk and the sliced walk at 640 x 360, and against the oracle on a 64 x 36 frame of
    the same camera."""
    wl = WL.make_workload("dragon_field_k8_1080p")
    assert len(wl.scene.tris()) == 64 * 18520
    tracer.upload(wl.scene)
    tracer.set_camera(wl.camera)
    out = {}
    for trav in (R.TRAV_FAST, R.TRAV_FAST_REFILL, R.TRAV_FAST_SLICED, R.TRAV_LANE):
        tracer.set_options(bounce_limit=8, seed=3, collect_stats=True, traversal=trav)
        out[trav] = tracer.trace(640, 360, 1, 2, want_primary=True)
    fi, fp, fs = out[R.TRAV_FAST]
    assert (fp != R.MISS).mean() > 0.3
    for trav in (R.TRAV_FAST_REFILL, R.TRAV_FAST_SLICED, R.TRAV_LANE):
        oi, op, os2 = out[trav]
        assert (os2.rays, os2.tri_hits) == (fs.rays, fs.tri_hits), trav
        assert np.array_equal(op, fp) and np.array_equal(bits(oi), bits(fi)), trav
    assert (out[R.TRAV_FAST_REFILL][2].box_tests, out[R.TRAV_FAST_REFILL][2].tri_tests) == (fs.box_tests, fs.tri_tests)
    osc = oracle_workload_scene(wl)
    opt = O.make_options(bounce_limit=8, seed=3, trig=O.TRIG_POLY, threads=16)
    oimg, oprim, ost = osc.render(oracle_camera(wl.camera), opt, 64, 36, 1, 2, want_primary=True)
    for trav in (R.TRAV_FAST, R.TRAV_LANE):
        tracer.set_options(bounce_limit=8, seed=3, collect_stats=True, traversal=trav)
        img, prim, st = tracer.trace(64, 36, 1, 2, want_primary=True)
        assert np.array_equal(prim, oprim) and np.array_equal(bits(img), bits(oimg)), trav
        assert (st.rays, st.tri_hits) == (ost.rays, ost.tri_hits)
        if trav == R.TRAV_LANE:
            assert (st.box_tests, st.tri_tests) == (ost.box_tests, ost.tri_tests)


def test_ten_million_triangle_field(tracer):
    """Configuration 5's large field: 23 x 23 dragons, 9 797 080 triangles in one flat list -- nine times the reference's own
    capacity of 2^20 triangles (CPURayTrace/SimplifiedData.hpp:90) and deeper than its stack of 32 (SimplifiedData.cpp:432),
    neither of which this build inherits.  The default walk against the per-lane reference-order walk at 480 x 270 (ids, ray
    count, every float of the image), against the oracle on a 32 x 18 frame of the same camera, and the configuration's
    sample-range split: two shares of the samples, each with its own sample offset, sum to the whole (to rounding: the mean
    of two running means is not the running mean)."""
    wl = WL.make_workload("dragon_field_k23_1080p")
    assert len(wl.scene.tris()) == 23 * 23 * 18520 > 1 << 20
    tracer.upload(wl.scene)
    tracer.set_camera(wl.camera)
    out = {}
    for trav in (R.TRAV_FAST, R.TRAV_FAST_SLICED, R.TRAV_LANE):
        tracer.set_options(bounce_limit=wl.bounce_limit, seed=5, collect_stats=(trav == R.TRAV_LANE), traversal=trav)
        out[trav] = tracer.trace(480, 270, 1, 2, want_primary=True)
    fi, fp, fs = out[R.TRAV_FAST]
    assert (fp != R.MISS).mean() > 0.3 and len(np.unique(fp // 18520)) > 200   # most of the 529 dragons are seen
    for trav in (R.TRAV_FAST_SLICED, R.TRAV_LANE):
        oi, op, os2 = out[trav]
        assert os2.rays == fs.rays and np.array_equal(op, fp) and np.array_equal(bits(oi), bits(fi)), trav
    _image_properties(fi)
    # the sample-range split of the configuration: shares [1, 2] and [3, 4] of four samples, RNG streams by sample offset
    tracer.set_options(bounce_limit=wl.bounce_limit, seed=5)
    whole, _, wst = tracer.trace(480, 270, 1, 4)
    parts, rays = [], 0
    for share in range(2):
        tracer.set_options(bounce_limit=wl.bounce_limit, seed=5, sample_offset=2 * share)
        part, _, pst = tracer.trace(480, 270, 1, 2)
        parts.append(part)
        rays += pst.rays
    assert rays == wst.rays
    assert np.array_equal(bits(parts[0]), bits(fi))
    assert np.allclose(0.5 * (parts[0] + parts[1]), whole, rtol=1e-5, atol=1e-6)
    # the oracle: its own builder over the 9.8 M triangles, the reference's traversal
    osc = oracle_workload_scene(wl)
    opt = O.make_options(bounce_limit=wl.bounce_limit, seed=5, trig=O.TRIG_POLY, threads=16)
    oimg, oprim, ost = osc.render(oracle_camera(wl.camera), opt, 32, 18, 1, 2, want_primary=True)
    assert (oprim != R.MISS).any()
    for trav in (R.TRAV_FAST, R.TRAV_LANE):
        tracer.set_options(bounce_limit=wl.bounce_limit, seed=5, collect_stats=True, traversal=trav)
        img, prim, st = tracer.trace(32, 18, 1, 2, want_primary=True)
        assert np.array_equal(prim, oprim) and np.array_equal(bits(img), bits(oimg)), trav
        assert (st.rays, st.tri_hits) == (ost.rays, ost.tri_hits), trav
        if trav == R.TRAV_LANE:
            assert (st.box_tests, st.tri_tests) == (ost.box_tests, ost.tri_tests)


def test_config4_as_stated_4k_256_samples(tracer, config4):
    """Configuration 4 exactly as stated: 3840 x 2160, 256 samples per pixel, bounce limit 8 (about 6 G rays, eleven chunks),
    one walk (the default).  No other implementation finishes this size in a test's time, so the checks are properties:
    the ray count is that of sixteen calls' worth of the 16-sample run scaled (within the sampling noise of the paths'
    lengths), the image is finite, non-negative, alpha in [0, 1], and equals -- bit for bit -- the same 256 samples traced
    as 200 + 56 in two progressive calls."""
    wl = config4
    tracer.upload(wl.scene)
    tracer.set_camera(wl.camera)
    tracer.set_options(bounce_limit=wl.bounce_limit, seed=wl.seed, jitter=wl.jitter)
    img, prim, st = tracer.trace(wl.width, wl.height, 1, 256, want_primary=True)
    _image_properties(img)
    assert (prim != R.MISS).mean() > 0.5
    n = wl.width * wl.height * 256
    assert 1.5 * n < st.rays < (wl.bounce_limit + 0.0) * n
    tracer.trace(wl.width, wl.height, 1, 200, want_image=False)
    img2, _, st2 = tracer.trace(wl.width, wl.height, 201, 56)
    assert np.array_equal(bits(img), bits(img2))


ALL_TRAVS = [R.TRAV_FAST, R.TRAV_FAST_REPLAY, R.TRAV_FAST_REFILL, R.TRAV_FAST_QUEUED, R.TRAV_FAST_INLINE, R.TRAV_FAST_SLICED,
             R.TRAV_LANE, R.TRAV_WAVE, R.TRAV_NESTED]


def test_flat_scene_in_the_recursive_traversal_order(tracer):
    """sd::BVH::Intersect (SimplifiedData.cpp:401-428) visits the LEFT child first, IntersectLoop the right one; with
    every triangle present twice each hit is a tie that the two orders resolve differently.  The flat scene uploaded with
    RT_ORDER_LEFT_FIRST must reproduce the oracle's recursive traversal, ids, counts and image."""
    tris, _ = objparse.load_obj(WL.DRAGON_2426)
    perm = np.random.default_rng(2).permutation(len(tris))
    soup = np.concatenate([tris, tris[perm]])
    ps = R.SdScene().add_model(soup).finalize()
    os_ = O.SdScene()
    os_.add_model(soup, [len(soup)])
    os_.finalize()
    cam = WL.DRAGON_2426_CAMERA
    res = {}
    for order, otrav in ((R.ORDER_LEFT_FIRST, O.TRAV_RECURSIVE), (R.ORDER_RIGHT_FIRST, O.TRAV_LOOP)):
        opt = O.make_options(bounce_limit=6, seed=8, trig=O.TRIG_POLY, traversal=otrav)
        oimg, oprim, ost = os_.render(oracle_camera(cam), opt, 256, 144, 1, 2, want_primary=True)
        tracer.upload_arrays(ps.nodes(), ps.root, ps.tris(), np.zeros(0), np.zeros(0), order)
        tracer.set_camera(cam)
        for trav in ALL_TRAVS:
            tracer.set_options(bounce_limit=6, seed=8, collect_stats=True, traversal=trav)
            img, prim, st = tracer.trace(256, 144, 1, 2, want_primary=True)
            assert np.array_equal(prim, oprim), (order, trav)
            assert (st.rays, st.tri_hits) == (ost.rays, ost.tri_hits), (order, trav)
            if trav in (R.TRAV_LANE, R.TRAV_WAVE, R.TRAV_NESTED):
                assert (st.box_tests, st.tri_tests) == (ost.box_tests, ost.tri_tests), (order, trav)
            assert np.array_equal(bits(img), bits(oimg)), (order, trav)
        res[order] = oprim
    assert np.count_nonzero(res[R.ORDER_LEFT_FIRST] != res[R.ORDER_RIGHT_FIRST]) > 100   # the order does decide the ties


def test_wave_walked_camera_batches_against_the_oracle(tracer):
    """The 2 426-triangle dragon on a ground quad at 1280 x 720 x 14 spp: 12.9 M pixel samples, more than the re-fill kernel
    takes in one launch, so the first round runs in k_sliced, whose batches of fresh camera rays -- generated by the kernel
    itself, the frame being all scene -- walk the tree wave-synchronously.  Ids, ray counts and every float of the image equal
    the oracle's, with statistics (exact records throughout) and without (the rounds after the first step through the grid
    records).  (Camera batches queued by the camera stage take the same walk once a round holds more than 12 Mi of them --
    a frame of the headline's size; test_gpu_compact.py holds that path to the same bits under a lowered threshold.)"""
    tris, sizes = objparse.load_obj(WL.DRAGON_2426)
    W, H, S = 1280, 720, 14
    cam = R.Camera(1.0, (0.05, 0.22, 0.45), 2.0, 16.0 / 9.0, (-0.006, 0.06, -0.014))
    quad = WL.ground_quad(y=0.0, x0=-2.0, x1=2.0, z0=-2.0, z1=2.0)
    sc, osc = R.SdScene(), O.SdScene()
    sc.add_model(tris)
    osc.add_model(tris, sizes)
    sc.add_model(quad)
    osc.add_model(quad, [len(quad)])
    sc.finalize()
    osc.finalize()
    opt = O.make_options(bounce_limit=6, seed=21, jitter=0.001, trig=O.TRIG_POLY, threads=16)
    oimg, oprim, ost = osc.render(oracle_camera(cam), opt, W, H, 1, S, want_primary=True)
    assert (oprim != R.MISS).sum() > 20000
    tracer.upload(sc)
    tracer.set_camera(cam)
    for stats in (True, False):
        tracer.set_options(bounce_limit=6, seed=21, jitter=0.001, collect_stats=stats)
        img, prim, st = tracer.trace(W, H, 1, S, want_primary=True)
        assert np.array_equal(prim, oprim), stats
        assert st.rays == ost.rays, stats
        assert np.array_equal(bits(img), bits(oimg)), stats


FULL_SIZE = r"""
import hashlib, sys
sys.path.insert(0, sys.argv[1])
import raytracing_amd as R
from raytracing_amd import workloads as WL
for name, spp in (("dragon18520_1080p", 16), ("dragon18520_full_ground_1080p", 4), ("dragon_field_k8_1080p", 4)):
    wl = WL.make_workload(name)
    t = R.Tracer(0)
    t.upload(wl.scene)
    t.set_camera(wl.camera)
    t.set_options(bounce_limit=wl.bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter)
    img, prim, st = t.trace(wl.width, wl.height, 1, spp, want_primary=True)
    print(name, st.rays, st.replays, hashlib.sha256(img.tobytes()).hexdigest(), hashlib.sha256(prim.tobytes()).hexdigest())
    t.close()
"""


def test_wide_walk_and_its_distance_limit_at_full_size():
    """The bench workloads at 1920x1080 -- 33 M to 133 M camera rays and their bounces each -- under the default walk (wide
    records, distance limit, camera stage with its tile mask), without the limit, without the wide records (the grid records'
    walk) and without the tile mask: the same image and
    the same primary ids, byte for byte.  Each setting in a process of its own (the switches are read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({}, {"RT_PRUNE": "0"}, {"RT_WIDE": "0"}, {"RT_TILE_MASK": "0"}):
        r = subprocess.run([sys.executable, "-c", FULL_SIZE, root], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (env, r.stderr[-2000:])
        outs.append(r.stdout.strip().splitlines())
    assert len(outs[0]) == 3
    assert outs[1] == outs[0] and outs[2] == outs[0] and outs[3] == outs[0]
