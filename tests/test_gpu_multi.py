"""Several devices behind one context (rt_create_multi), through the C ABI.  Where the machine shows as many GPUs as a
test asks for, the device list names distinct ones (peer copies cross xGMI); on a box with one GPU it names that one two
or three times: the host threads, the per-device shares, the peer copies and the assembly run as they would over distinct
GPUs."""
import numpy as np
import pytest

from helpers import R, WL, bits, product_sd_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    return product_sd_scene([WL.DRAGON_2426])


def device_list(n):
    """n distinct HIP devices when the machine has them, device 0 n times otherwise."""
    import torch
    return list(range(n)) if torch.cuda.device_count() >= n else [0] * n


@pytest.mark.parametrize("n_dev", [2, 3])
def test_tile_split_is_bit_identical_to_one_device(scene, n_dev):
    one = R.Tracer(0)
    one.upload(scene)
    one.set_camera(WL.DRAGON_2426_CAMERA)
    one.set_options(bounce_limit=6, seed=4, collect_stats=True)
    multi = R.Tracer(devices=device_list(n_dev))
    assert multi.device_count() == n_dev
    multi.upload(scene)
    multi.set_camera(WL.DRAGON_2426_CAMERA)
    multi.set_options(bounce_limit=6, seed=4, collect_stats=True, multi_split=R.SPLIT_TILE)
    for (w, h) in ((250, 141), (640, 360)):
        a, ap, ast = one.trace(w, h, 1, 3, want_primary=True)
        b, bp, bst = multi.trace(w, h, 1, 3, want_primary=True)
        assert np.array_equal(ap, bp) and np.array_equal(bits(a), bits(b))
        assert (ast.rays, ast.box_tests, ast.tri_tests, ast.tri_hits) == (bst.rays, bst.box_tests, bst.tri_tests, bst.tri_hits)
        a, _, _ = one.trace(w, h, 4, 2)      # the progressive protocol goes on across calls
        b, _, _ = multi.trace(w, h, 4, 2)
        assert np.array_equal(bits(a), bits(b))
    one.close()
    multi.close()


def test_sample_split_matches_one_device_to_rounding(scene):
    one = R.Tracer(0)
    one.upload(scene)
    one.set_camera(WL.DRAGON_2426_CAMERA)
    one.set_options(bounce_limit=6, seed=4)
    multi = R.Tracer(devices=device_list(3))
    multi.upload(scene)
    multi.set_camera(WL.DRAGON_2426_CAMERA)
    multi.set_options(bounce_limit=6, seed=4, multi_split=R.SPLIT_SAMPLE)
    a, ap, ast = one.trace(320, 180, 1, 8, want_primary=True)
    b, bp, bst = multi.trace(320, 180, 1, 8, want_primary=True)   # shares 3, 3, 2
    assert ast.rays == bst.rays and np.array_equal(ap, bp)
    # a weighted sum of three running means against one running mean: rounding only
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
    a, _, ast = one.trace(320, 180, 9, 4)
    b, _, bst = multi.trace(320, 180, 9, 4)                      # continues: shares 2, 1, 1
    assert ast.rays == bst.rays
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
    # the devices' means belong to the context: a call that neither restarts them nor continues this very run is refused
    with pytest.raises(R.TracerError) as e:
        multi.trace(320, 180, 20, 2)          # samples 13 .. 19 were never rendered here
    assert e.value.code == -1 and "first_sample = 13" in str(e.value)
    with pytest.raises(R.TracerError):
        multi.trace(160, 90, 13, 2)           # another frame size without a restart
    b, _, _ = multi.trace(320, 180, 13, 1)    # the run itself goes on
    a, _, _ = one.trace(320, 180, 13, 1)
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
    one.close()
    multi.close()


def test_multi_device_output_on_the_device_and_refusals(scene):
    import torch
    multi = R.Tracer(devices=device_list(2))
    multi.upload(scene)
    multi.set_camera(WL.DRAGON_2426_CAMERA)
    multi.set_options(bounce_limit=4, seed=9)
    multi.reserve(256, 144, 2)
    img = torch.zeros((144, 256, 4), dtype=torch.float32, device="cuda:0")
    multi.trace_device(256, 144, 1, 2, img.data_ptr())
    multi.sync()
    ref, _, _ = multi.trace(256, 144, 1, 2)
    assert np.array_equal(bits(img.cpu().numpy()), bits(ref))
    with pytest.raises(R.TracerError):
        multi.set_pipelining(True)
    multi.close()
    with pytest.raises(R.TracerError) as e:
        R.Tracer(devices=[0, 99])
    assert e.value.code == -2


def test_frames_begun_ahead_on_two_devices(scene):
    """rt_trace_begin / rt_trace_end on a context over two devices (tile split): the frames of one device's synchronous
    calls, bit for bit, two always in flight; a discard leaves no trace; the sample split refuses."""
    import ctypes as C
    W, H = 320, 180
    one = R.Tracer(0)
    one.upload(scene)
    one.set_camera(WL.DRAGON_2426_CAMERA)
    one.set_options(bounce_limit=6, seed=4)
    frames = [one.trace(W, H, n, 1)[0].copy() for n in range(1, 8)]
    one.close()
    multi = R.Tracer(devices=device_list(2))
    multi.upload(scene)
    multi.set_camera(WL.DRAGON_2426_CAMERA)
    multi.set_options(bounce_limit=6, seed=4)
    bufs = []
    for _ in range(3):
        p = R.Tracer.alloc_host(W * H * 16)
        bufs.append((p, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(H, W, 4))))
    multi.trace_begin(W, H, 1, 1, bufs[0][0])
    multi.trace_begin(W, H, 2, 1, bufs[1][0])
    with pytest.raises(R.TracerError):
        multi.trace_begin(W, H, 3, 1, bufs[2][0])
    for n in range(1, 5):
        st = multi.trace_end()
        assert st.rays > 0
        assert np.array_equal(bits(bufs[(n - 1) % 3][1]), bits(frames[n - 1])), n
        multi.trace_begin(W, H, n + 2, 1, bufs[(n + 1) % 3][0])
    multi.trace_discard()                       # frames 5 and 6 were in flight
    img, _, _ = multi.trace(W, H, 5, 1)         # the synchronous call continues from frame 4
    assert np.array_equal(bits(img), bits(frames[4]))
    multi.trace_begin(W, H, 6, 2, bufs[0][0])   # two samples in one call
    multi.trace_end()
    assert np.array_equal(bits(bufs[0][1]), bits(frames[6]))
    multi.set_options(bounce_limit=6, seed=4, multi_split=R.SPLIT_SAMPLE)
    with pytest.raises(R.TracerError):
        multi.trace_begin(W, H, 1, 2, bufs[1][0])
    for p, _ in bufs:
        R.Tracer.free_host(p)
    multi.close()
