"""GPU parity on edge cases, through the C ABI, every traversal against the oracle, bit for bit:
equal-distance ties (duplicated geometry: the reference keeps the first hit in ITS visiting order), rays with an
exactly zero direction component whose origin lies exactly on box planes (0 * inf = NaN in the slab test,
SimplifiedData.cpp:146-161), degenerate and sliver triangles, one- and two-leaf trees, tiny and ragged images,
bounce limits 0 and 1."""
import numpy as np
import pytest

from helpers import O, R, WL, bits, objparse, oracle_camera

pytestmark = pytest.mark.gpu

TRAVS = [R.TRAV_FAST, R.TRAV_FAST_REPLAY, R.TRAV_FAST_REFILL, R.TRAV_FAST_QUEUED, R.TRAV_FAST_INLINE, R.TRAV_FAST_SLICED, R.TRAV_LANE, R.TRAV_WAVE, R.TRAV_NESTED]


@pytest.fixture(scope="module")
def tracer():
    t = R.Tracer(0)
    yield t
    t.close()


def both_scenes(models):
    ps, os_ = R.SdScene(), O.SdScene()
    for tris in models:
        t = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 24)
        ps.add_model(t)
        os_.add_model(t, [len(t)])
    ps.finalize()
    os_.finalize()
    return ps, os_


def compare(tracer, ps, os_, cam, w, h, spp=2, bounce=8, seed=3, jitter=0.001, min_hits=1):
    tracer.upload(ps)
    tracer.set_camera(cam)
    opt = O.make_options(bounce_limit=bounce, seed=seed, jitter=jitter, trig=O.TRIG_POLY)
    oimg, oprim, ost = os_.render(oracle_camera(cam), opt, w, h, 1, spp, want_primary=True)
    assert (oprim != R.MISS).sum() >= min_hits
    for trav in TRAVS:
        tracer.set_options(bounce_limit=bounce, seed=seed, jitter=jitter, collect_stats=True, traversal=trav)
        img, prim, st = tracer.trace(w, h, 1, spp, want_primary=True)
        assert np.count_nonzero(prim != oprim) == 0, trav
        assert st.rays == ost.rays, trav
        assert np.array_equal(bits(img), bits(oimg)), trav
    return oprim


def tri(p0, p1, p2, n=(0.0, 0.0, 1.0)):
    return np.array([[*p0, *n, 0, 0], [*p1, *n, 0, 0], [*p2, *n, 0, 0]], dtype=np.float32).reshape(24)


def test_equal_distance_ties_duplicated_geometry(tracer):
    tris, _ = objparse.load_obj(WL.DRAGON_2426)
    ps, os_ = both_scenes([tris, tris.copy(), tris.copy()])   # three coincident meshes: every hit is a three-way tie
    compare(tracer, ps, os_, WL.DRAGON_2426_CAMERA, 320, 180, min_hits=2000)
    # one mesh that holds every triangle twice, the second copies shuffled: which copy wins depends on where the
    # reference's sort and median splits put them, and it is not always the same copy
    perm = np.random.default_rng(1).permutation(len(tris))
    ps, os_ = both_scenes([np.concatenate([tris, tris[perm]])])
    prim = compare(tracer, ps, os_, WL.DRAGON_2426_CAMERA, 320, 180, min_hits=2000)
    hit = prim[prim != R.MISS]
    assert len(np.unique(hit // len(tris))) == 2


def test_zero_direction_component_on_box_planes(tracer):
    """Camera looks straight down -z; the middle pixel column has dir.x == 0 exactly, and the camera's x equals the
    leaf boxes' pMin.x (vertex x minus the 1e-5 bias, in fp32), so (pMin.x - o.x) * (1 / 0) = 0 * inf = NaN."""
    vx = np.float32(0.5)
    on_plane = float(np.float32(vx - np.float32(1e-5)))
    quads = []
    for k, z in enumerate((0.0, -0.5, -1.0)):
        quads.append(tri((vx, -1, z), (vx + 1, -1, z), (vx + 1, 1, z)))
        quads.append(tri((vx, -1, z), (vx + 1, 1, z), (vx, 1, z)))
        quads.append(tri((vx - 1, -1, z - 0.1), (vx, -1, z - 0.1), (vx, 1, z - 0.1)))
    ps, os_ = both_scenes([np.stack(quads)])
    cam = R.Camera(1.0, (on_plane, 0.0, 1.0), 2.0, 16.0 / 9.0, (on_plane, 0.0, 0.0))
    compare(tracer, ps, os_, cam, 64, 36, spp=1, jitter=0.0, min_hits=100)
    compare(tracer, ps, os_, cam, 64, 36, spp=3, jitter=0.001, min_hits=100)
    cam_y = R.Camera(1.0, (1.0, 0.0, 1.0), 2.0, 16.0 / 9.0, (1.0, 0.0, 0.0))  # dir.y == 0 on the middle row as well
    compare(tracer, ps, os_, cam_y, 64, 36, spp=1, jitter=0.0, min_hits=100)


def test_degenerate_and_sliver_triangles(tracer):
    rng = np.random.default_rng(5)
    soup = []
    for i in range(300):
        c = rng.uniform(-1, 1, 3).astype(np.float32)
        a, b = rng.normal(0, 0.3, 3).astype(np.float32), rng.normal(0, 0.3, 3).astype(np.float32)
        kind = i % 6
        if kind == 0:      # zero area: repeated vertex
            soup.append(tri(c, c, c + a))
        elif kind == 1:    # zero area: collinear
            soup.append(tri(c, c + a, c + a * np.float32(2)))
        elif kind == 2:    # sliver
            soup.append(tri(c, c + a, c + a + b * np.float32(1e-6)))
        else:
            nrm = np.cross(a, b)
            nrm = nrm / (np.linalg.norm(nrm) + 1e-20)
            soup.append(tri(c, c + a, c + b, nrm))
    ps, os_ = both_scenes([np.stack(soup)])
    cam = R.Camera(1.0, (0.0, 0.0, 3.5), 2.0, 16.0 / 9.0, (0.0, 0.0, 0.0))
    compare(tracer, ps, os_, cam, 200, 113, spp=2, min_hits=1000)
    compare(tracer, ps, os_, R.Camera(1.0, (0.1, 0.05, 0.2), 2.0, 16.0 / 9.0, (0.0, 0.3, -1.0)), 160, 90, spp=2, min_hits=500)  # from inside


@pytest.mark.parametrize("n", [1, 2, 3])
def test_smallest_trees(tracer, n):
    tris = [tri((-1, -1, -0.5 * k), (1, -1, -0.5 * k), (0, 1, -0.5 * k)) for k in range(n)]
    ps, os_ = both_scenes([np.stack(tris)])
    compare(tracer, ps, os_, R.Camera(1.0, (0.0, 0.0, 2.0), 2.0, 16.0 / 9.0, (0.0, 0.0, 0.0)), 96, 54, min_hits=200)


@pytest.mark.parametrize("size", [(1, 1), (7, 3), (9, 17), (65, 8)])
def test_tiny_and_ragged_images(tracer, size):
    tris, _ = objparse.load_obj(WL.DRAGON_2426)
    ps, os_ = both_scenes([tris])
    compare(tracer, ps, os_, WL.DRAGON_2426_CAMERA, size[0], size[1], spp=5, min_hits=0)


@pytest.mark.parametrize("bounce", [0, 1, 2])
def test_small_bounce_limits(tracer, bounce):
    tris, _ = objparse.load_obj(WL.DRAGON_2426)
    ps, os_ = both_scenes([tris])
    compare(tracer, ps, os_, WL.DRAGON_2426_CAMERA, 160, 90, spp=3, bounce=bounce, min_hits=0 if bounce == 0 else 500)


def test_many_samples_cross_chunks(tracer):
    """More samples than one wavefront chunk holds (16), starting from a sample index > 1."""
    tris, _ = objparse.load_obj(WL.DRAGON_2426)
    ps, os_ = both_scenes([tris])
    tracer.upload(ps)
    tracer.set_camera(WL.DRAGON_2426_CAMERA)
    tracer.set_options(bounce_limit=6, seed=8)
    a, _, _ = tracer.trace(128, 72, 1, 5)
    b, _, _ = tracer.trace(128, 72, 6, 40)
    opt = O.make_options(bounce_limit=6, seed=8, trig=O.TRIG_POLY)
    oimg, _, _ = os_.render(oracle_camera(WL.DRAGON_2426_CAMERA), opt, 128, 72, 1, 45)
    assert np.array_equal(bits(b), bits(oimg))


def test_more_samples_than_one_chunk_holds(tracer):
    """130 samples of a small image = three chunks (64 + 64 + 2), which the tracer runs alternately on its two
    internal streams; the folds into the running mean must stay in sample order."""
    tris, _ = objparse.load_obj(WL.CYLINDER)
    ps, os_ = both_scenes([tris])
    cam = R.Camera(1.0, (0.5, 1.5, 6.0), 2.0, 16.0 / 9.0, (0.0, 0.8, 0.0))
    tracer.upload(ps)
    tracer.set_camera(cam)
    tracer.set_options(bounce_limit=4, seed=12)
    img, _, _ = tracer.trace(96, 54, 1, 130)
    oimg, _, _ = os_.render(oracle_camera(cam), O.make_options(bounce_limit=4, seed=12, trig=O.TRIG_POLY), 96, 54, 1, 130)
    assert np.array_equal(bits(img), bits(oimg))


def test_pipelined_device_calls(tracer):
    """rt_set_pipelining: consecutive rt_trace_device calls overlap on the tracer's own streams; rt_join / rt_sync
    order the result.  Seven progressive calls of 3 samples each into a caller-owned device image must give the
    21-sample running mean, bit for bit, and the call after switching pipelining off again must see it too."""
    torch = pytest.importorskip("torch")
    tris, _ = objparse.load_obj(WL.DRAGON_2426)
    ps, os_ = both_scenes([tris])
    t = R.Tracer(0)
    t.upload(ps)
    t.set_camera(WL.DRAGON_2426_CAMERA)
    t.set_options(bounce_limit=8, seed=21)
    w, h = 200, 113
    img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    t.set_pipelining(True)
    for i in range(7):
        t.trace_device(w, h, 1 + 3 * i, 3, img.data_ptr())
    t.join()
    st = t.sync()
    assert st.launches == 7
    opt = O.make_options(bounce_limit=8, seed=21, trig=O.TRIG_POLY)
    oimg, _, ost = os_.render(oracle_camera(WL.DRAGON_2426_CAMERA), opt, w, h, 1, 21)
    assert st.rays == ost.rays
    assert np.array_equal(bits(img.cpu().numpy()), bits(oimg))
    t.set_pipelining(False)
    t.trace_device(w, h, 22, 2, img.data_ptr())
    t.sync()
    oimg2, _, _ = os_.render(oracle_camera(WL.DRAGON_2426_CAMERA), opt, w, h, 1, 23)
    assert np.array_equal(bits(img.cpu().numpy()), bits(oimg2))
    t.close()


def test_camera_a_hair_above_the_surface(tracer):
    """Camera half a micrometre above a vertex shared by several triangles, looking at it: the primary hits lie
    between the triangle test's 1e-7 and the slab test's 1e-6 lower bound, in front of their own leaf boxes' entry
    distance, so the fast walk has to replay the reference's acceptance rule in visiting order for most hitting rays."""
    tris, _ = objparse.load_obj(WL.DRAGON_2426)
    ps, os_ = both_scenes([tris])
    t = tris.reshape(-1, 3, 8)
    replays = 0
    for k in (17, 400, 1300):
        v, n = t[k, 0, 0:3].astype(np.float32), t[k, 0, 3:6].astype(np.float32)
        n = n / np.linalg.norm(n)
        pos = (v + n * np.float32(5e-7)).astype(np.float32)
        cam = R.Camera(1.0, tuple(float(x) for x in pos), 2.0, 16.0 / 9.0, tuple(float(x) for x in v - n))
        compare(tracer, ps, os_, cam, 64, 36, spp=2, jitter=0.001, min_hits=50)
        tracer.set_options(bounce_limit=8, seed=3, jitter=0.001, traversal=R.TRAV_FAST)
        _, _, st = tracer.trace(64, 36, 1, 2)
        replays += st.replays
    assert replays > 100


def test_more_noted_tests_than_a_wave_can_hold(tracer):
    """Forty big triangles stacked behind each other, each a leaf whose box every camera ray passes: a wave of 64 rays
    meets 2 560 leaves per search, five times what its list of noted primitive tests holds (512), so most tests take the
    run-on-the-spot path of the queued and re-fill walks while the rest go through the list -- same bits either way.  A
    second model of small triangles in front makes the closest hit come sometimes from the list, sometimes from the spot."""
    rng = np.random.default_rng(11)
    big = []
    for k in range(40):
        z = -2.0 - 0.05 * k
        big.append(tri((-6.0, -4.0, z), (6.0, -4.0, z + 0.01 * (k % 3)), (0.0, 6.0, z)))
    small = []
    for _ in range(300):
        c = rng.uniform(-1.5, 1.5, size=3) * np.array([1.0, 0.7, 0.3]) + np.array([0.0, 0.0, -1.2])
        a, b2 = rng.normal(size=3) * 0.12, rng.normal(size=3) * 0.12
        nrm = np.cross(a, b2)
        nrm = nrm / (np.linalg.norm(nrm) + 1e-12)
        small.append(tri(c, c + a, c + b2, tuple(nrm)))
    ps, os_ = both_scenes([np.array(big), np.array(small)])
    cam = R.Camera(1.0, (0.0, 0.0, 1.0), 2.0, 16.0 / 9.0, (0.0, 0.0, -2.0))
    prim = compare(tracer, ps, os_, cam, 160, 90, spp=3, min_hits=160 * 90 // 2)
    assert len(np.unique(prim)) > 40


def test_random_soups_every_fast_variant_against_the_oracle(tracer):
    """Random triangle soups (dense overlap: long candidate lists, many bounces between nearby triangles), random cameras:
    every traversal against the oracle, bit for bit."""
    rng = np.random.default_rng(2024)
    for case in range(4):
        n = (60, 400, 1500, 3000)[case]
        c = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32) * np.array([1.0, 0.6, 1.0], dtype=np.float32)
        a = rng.normal(size=(n, 3)).astype(np.float32) * (0.5 if case < 2 else 0.15)
        b2 = rng.normal(size=(n, 3)).astype(np.float32) * (0.5 if case < 2 else 0.15)
        nrm = np.cross(a, b2)
        nrm /= (np.linalg.norm(nrm, axis=1, keepdims=True) + 1e-12)
        tris = np.zeros((n, 3, 8), dtype=np.float32)
        tris[:, 0, :3], tris[:, 1, :3], tris[:, 2, :3] = c, c + a, c + b2
        tris[:, :, 3:6] = nrm[:, None, :]
        ps, os_ = both_scenes([tris.reshape(n, 24)])
        eye = tuple(float(v) for v in rng.uniform(-0.3, 0.3, size=3) + np.array([0.0, 0.0, 2.5 - case * 0.6]))
        cam = R.Camera(1.0, eye, 2.0, 16.0 / 9.0, (0.0, 0.0, 0.0))
        compare(tracer, ps, os_, cam, 128, 72, spp=2, seed=case + 1, min_hits=128 * 72 // 4)


def test_load_and_trace_from_different_threads(tracer):
    """The interface demands that load() and trace() may come from any thread (Renderer/RenderInterfaces.hpp:4-10, 14-20;
    the reference takes a unique_lock in load and a shared_lock in trace): one thread keeps replacing the scene while
    another keeps tracing; every frame must be exactly the frame of one scene or of the other, never a mixture."""
    import threading
    tris_a, _ = objparse.load_obj(WL.DRAGON_2426)
    tris_b, _ = objparse.load_obj(WL.CYLINDER)
    scene_a, _ = both_scenes([tris_a])
    scene_b, _ = both_scenes([tris_b * np.float32(0.05) + np.float32(0.0)])
    cam = WL.DRAGON_2426_CAMERA
    tracer.set_camera(cam)
    tracer.set_options(bounce_limit=4, seed=9, traversal=R.TRAV_FAST)
    expected = []
    for sc in (scene_a, scene_b):
        tracer.upload(sc)
        img, _, _ = tracer.trace(96, 54, 1, 2)
        expected.append(bits(img).copy())
    assert not np.array_equal(expected[0], expected[1])
    errors, seen = [], set()

    def loader():
        try:
            for i in range(30):
                tracer.upload(scene_a if i % 2 == 0 else scene_b)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def render():
        try:
            for _ in range(60):
                img, _, _ = tracer.trace(96, 54, 1, 2)
                b = bits(img)
                which = 0 if np.array_equal(b, expected[0]) else 1 if np.array_equal(b, expected[1]) else -1
                seen.add(which)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=loader), threading.Thread(target=render)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    assert -1 not in seen and seen
