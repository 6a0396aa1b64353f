"""The compact-tree walk (16-byte grid records of a flat scene, taken by traces that collect no statistics) against the oracle
and against the walk over the exact records (which a trace with statistics takes): same bits -- images, primary ids, ray
counts, replays -- on a bare mesh, on a mesh over a ground quad a hundred times its size (coarse grid; rays that leave
the ground straight up run along a coordinate axis), on a scene of no thickness, from cameras inside and outside the
range the grid arithmetic is good for, over chunked calls and tile shares, and through the forced sliced / re-fill kernels.
One case runs in a child process that makes the statistics kernels take the compact walk too, to see that it is the
compact walk that ran: more box tests, the same triangle tests and hits.  Another runs the same render under tuning
settings far from the defaults (slice length, hand-over threshold, quorum, chunk size, extra rounds, each kernel of the
sliced walk finishing its own rounds): how the work is cut into launches must not show in the image."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import O, R, WL, bits, oracle_camera

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scenes(models):
    ps, os_ = R.SdScene(), O.SdScene()
    for m in models:
        ps.add_model(m)
        os_.add_model(m, [len(m)])
    os_.finalize()
    return ps.finalize(), os_


@pytest.fixture(scope="module")
def tracer():
    t = R.Tracer(0)
    yield t
    t.close()


@pytest.fixture(scope="module")
def dragon_tris():
    return WL._load_obj_tris(WL.DRAGON_2426)


def check(tracer, os_, cam, w, h, spp, bounce=8, seed=11, jitter=0.001, travs=(R.TRAV_AUTO, R.TRAV_FAST_SLICED, R.TRAV_FAST_REFILL)):
    tracer.set_camera(cam)
    oimg, oprim, ost = os_.render(oracle_camera(cam), O.make_options(bounce_limit=bounce, seed=seed, trig=O.TRIG_POLY, jitter=jitter), w, h, 1, spp,
                                  want_primary=True)
    tracer.set_options(bounce_limit=bounce, seed=seed, jitter=jitter, collect_stats=True)
    _, _, exact = tracer.trace(w, h, 1, spp)
    for trav in travs:
        tracer.set_options(bounce_limit=bounce, seed=seed, jitter=jitter, traversal=trav)
        img, prim, st = tracer.trace(w, h, 1, spp, want_primary=True)
        assert np.array_equal(prim, oprim), trav
        assert np.array_equal(bits(img), bits(oimg)), trav
        assert (st.rays, st.replays) == (ost.rays, exact.replays), trav
    return ost


def test_bare_mesh(tracer, dragon_tris):
    ps, os_ = scenes([dragon_tris])
    tracer.upload(ps)
    check(tracer, os_, WL.DRAGON_2426_CAMERA, 200, 112, 6)
    check(tracer, os_, R.Camera(1.0, (0.0, 0.10, 0.9), 2.0, 16.0 / 9.0, (-0.006, 0.073, -0.014)), 331, 187, 3, bounce=3)


def test_mesh_over_a_large_ground(tracer, dragon_tris):
    ps, os_ = scenes([dragon_tris, WL.ground_quad(y=0.0, x0=-20.0, x1=20.0, z0=-20.0, z1=20.0)])
    tracer.upload(ps)
    cam = R.Camera(1.0, (0.05, 0.22, 0.45), 2.0, 16.0 / 9.0, (-0.006, 0.06, -0.014))
    ost = check(tracer, os_, cam, 240, 135, 8)
    assert ost.rays > 1.5 * 240 * 135 * 8          # the ground fills the frame: most paths bounce
    # centre column and row of an odd frame without jitter: camera rays with a zero direction component
    check(tracer, os_, R.Camera(1.0, (0.0, 0.05, 0.5), 2.0, 16.0 / 9.0, (0.0, 0.05, 0.0)), 101, 57, 2, jitter=0.0, travs=(R.TRAV_AUTO,))


def test_scene_without_thickness_and_distant_cameras(tracer):
    quad = WL.ground_quad(y=0.0, x0=-1.0, x1=1.0, z0=-1.0, z1=1.0)
    tri = np.array([[-0.3, 0.0, -0.2, 0, 1, 0, 0, 0, 0.4, 0.0, -0.1, 0, 1, 0, 0, 0, 0.0, 0.0, 0.5, 0, 1, 0, 0, 0]], dtype=np.float32)
    ps, os_ = scenes([quad, tri])
    tracer.upload(ps)
    # the grid is a quarter of a unit thick here: 5 units up is inside the range of the grid arithmetic, 40 units is not
    for height in (0.7, 5.0, 40.0):
        cam = R.Camera(4.0 * height, (0.1, height, 0.3), 2.0, 16.0 / 9.0, (0.0, 0.0, 0.0))
        check(tracer, os_, cam, 128, 72, 3, bounce=4, travs=(R.TRAV_AUTO,))


def test_chunks_and_tile_shares(tracer, dragon_tris):
    ps, _ = scenes([dragon_tris, WL.ground_quad(y=0.0, x0=-2.0, x1=2.0, z0=-2.0, z1=2.0)])
    tracer.upload(ps)
    tracer.set_camera(R.Camera(1.0, (0.05, 0.22, 0.45), 2.0, 16.0 / 9.0, (-0.006, 0.06, -0.014)))
    tracer.set_options(bounce_limit=5, seed=5, collect_stats=True)
    exact, eprim, est = tracer.trace(250, 141, 1, 70, want_primary=True)      # 70 samples: two chunks (64 + 6)
    tracer.set_options(bounce_limit=5, seed=5)
    whole, wprim, st = tracer.trace(250, 141, 1, 70, want_primary=True)
    assert np.array_equal(bits(whole), bits(exact)) and np.array_equal(wprim, eprim) and st.rays == est.rays
    acc, rays = np.zeros_like(whole), 0
    for r in range(3):
        tracer.set_options(bounce_limit=5, seed=5, tile_rank=r, tile_world=3)
        part, _, s = tracer.trace(250, 141, 1, 70)
        acc += part
        rays += s.rays
    assert rays == st.rays and np.array_equal(bits(whole), bits(acc))
    tracer.set_options(bounce_limit=5, seed=5)
    for first, n in ((1, 1), (2, 3), (5, 64), (69, 2)):
        part, _, _ = tracer.trace(250, 141, first, n)
    assert np.array_equal(bits(whole), bits(part))


CHILD = r"""
import hashlib, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import raytracing_amd as R
from raytracing_amd import workloads as WL
t = R.Tracer(0)
sc = R.SdScene()
sc.add_model(WL._load_obj_tris(WL.DRAGON_2426))
t.upload(sc.finalize())
t.set_camera(WL.DRAGON_2426_CAMERA)
for trav in (R.TRAV_AUTO, R.TRAV_FAST_REFILL):
    t.set_options(bounce_limit=8, seed=3, collect_stats=True, traversal=trav)
    img, _, st = t.trace(320, 180, 1, 4)
    print(trav, st.rays, st.box_tests, st.tri_tests, st.tri_hits, st.replays, hashlib.sha256(img.tobytes()).hexdigest())
"""


def test_statistics_of_the_compact_walk_itself():
    out = {}
    for mode in ("0", "2"):
        r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=dict(os.environ, RT_COMPACT=mode), capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[mode] = [line.split() for line in r.stdout.strip().splitlines()]
    for exact, compact in zip(out["0"], out["2"]):
        trav, rays, box, tri, hits, replays, sha = exact
        assert compact[0] == trav and compact[1] == rays and compact[3:] == [tri, hits, replays, sha]
        assert int(box) < int(compact[2]) < 1.5 * int(box)      # grid boxes are looser, and each leaf's exact test is counted too


TUNED = r"""
import hashlib, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import raytracing_amd as R
from raytracing_amd import workloads as WL
t = R.Tracer(0)
# a mesh on a ground quad (the frame is all scene: the path kernels generate the camera rays themselves) and the bare mesh
# (most of the frame is sky: the camera stage queues the camera rays that meet the scene's box)
# ... and forty big triangles stacked behind each other, each a leaf whose box every camera ray meets: a wave's 64 rays note
# five times what its list holds, which the wave-synchronous walk has to empty on the way
def tri(p0, p1, p2):
    return [*p0, 0, 0, 1, 0, 0, *p1, 0, 0, 1, 0, 0, *p2, 0, 0, 1, 0, 0]
stack = np.array([tri((-6.0, -4.0, -2.0 - 0.05 * k), (6.0, -4.0, -2.0 - 0.05 * k + 0.01 * (k % 3)), (0.0, 6.0, -2.0 - 0.05 * k)) for k in range(40)],
                 dtype=np.float32)
for ground, cam in ((True, R.Camera(1.0, (0.05, 0.22, 0.45), 2.0, 16.0 / 9.0, (-0.006, 0.06, -0.014))), (False, WL.DRAGON_2426_CAMERA),
                    (None, R.Camera(1.0, (0.0, 0.0, 1.0), 2.0, 16.0 / 9.0, (0.0, 0.0, -2.0)))):
    sc = R.SdScene()
    if ground is None:
        sc.add_model(stack)
    else:
        sc.add_model(WL._load_obj_tris(WL.DRAGON_2426))
    if ground:
        sc.add_model(WL.ground_quad(y=0.0, x0=-2.0, x1=2.0, z0=-2.0, z1=2.0))
    t.upload(sc.finalize())
    t.set_camera(cam)
    for trav in (R.TRAV_AUTO, R.TRAV_FAST_REFILL):
        t.set_options(bounce_limit=8, seed=9, traversal=trav)
        img, prim, st = t.trace(480, 270, 1, 40, want_primary=True)
        print(ground, trav, st.rays, st.replays, hashlib.sha256(img.tobytes()).hexdigest(), hashlib.sha256(prim.tobytes()).hexdigest())
"""


def test_results_do_not_depend_on_the_tuning():
    settings = [{},
                {"RT_SLICE": "7", "RT_TAIL_RAYS": "3000", "RT_SLICE_QUORUM": "40", "RT_CHUNK_SAMPLES": "13"},
                {"RT_SLICE": "1000", "RT_TAIL_RAYS": "100000000", "RT_SLICE_QUORUM": "0", "RT_EXTRA_ROUNDS": "5"},
                {"RT_REFILL_TAIL": "0", "RT_SLICE": "33", "RT_TAIL_RAYS": "50000", "RT_CHUNK_SAMPLES": "64", "RT_EXTRA_ROUNDS": "-3"},
                {"RT_COMPACT": "0", "RT_TAIL_RAYS": "1", "RT_SLICE_QUORUM": "63"},
                # camera rays lane by lane instead of wave-synchronously, in rounds of the sliced walk
                {"RT_WAVE_CAMERA": "0", "RT_TAIL_RAYS": "3000"}, {"RT_WAVE_CAMERA": "0", "RT_REFILL_TAIL": "0"},
                # no wide tree (the grid records' sliced walk with its re-fill hand-over), the wide tree without its distance
                # limit, and with the camera rays of every scene through it
                {"RT_WIDE": "0"}, {"RT_WIDE": "0", "RT_TAIL_RAYS": "3000"}, {"RT_PRUNE": "0"}, {"RT_PRUNE": "0", "RT_WAVE_CAMERA": "0"},
                {"RT_WAVE_CAMERA": "0", "RT_CHUNK_SAMPLES": "7"}, {"RT_WIDE_BURST": "1"}, {"RT_WIDE_BURST": "9", "RT_WAVE_CAMERA": "0"},
                # the camera stage without its tile mask
                {"RT_TILE_MASK": "0"}]
    outs = []
    for env in settings:
        r = subprocess.run([sys.executable, "-c", TUNED, ROOT], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (env, r.stderr[-2000:])
        outs.append(r.stdout.strip().splitlines())
    assert len(outs[0]) == 6
    for env, out in zip(settings[1:], outs[1:]):
        assert out == outs[0], env
