"""Frames computed ahead of the caller (rt_trace_begin / rt_trace_end / rt_trace_discard, and the plug-in's trace method
on top of them) give the very bits of one synchronous rt_trace per frame, whatever is dropped on the way."""
import ctypes as C

import numpy as np
import pytest

from helpers import R, WL, bits, product_sd_scene

pytestmark = pytest.mark.gpu
W, H = 320, 180


def _host_image():
    p = R.Tracer.alloc_host(W * H * 16)
    return p, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(H, W, 4))


def test_begin_end_equals_synchronous_calls_and_discard_leaves_no_trace():
    ps = product_sd_scene([WL.DRAGON_2426])
    ref = R.Tracer(0)
    ref.upload(ps)
    ref.set_camera(WL.DRAGON_2426_CAMERA)
    ref.set_options(bounce_limit=6, seed=4)
    frames = [ref.trace(W, H, n, 1)[0].copy() for n in range(1, 9)]
    t = R.Tracer(0)
    t.upload(ps)
    t.set_camera(WL.DRAGON_2426_CAMERA)
    t.set_options(bounce_limit=6, seed=4)
    bufs = [_host_image() for _ in range(3)]
    t.trace_begin(W, H, 1, 1, bufs[0][0])
    t.trace_begin(W, H, 2, 1, bufs[1][0])
    with pytest.raises(R.TracerError):            # a third call in flight is refused
        t.trace_begin(W, H, 3, 1, bufs[2][0])
    for n in range(1, 6):                          # frames 1 .. 5, two always in flight
        st = t.trace_end()
        assert st.rays > 0
        assert np.array_equal(bits(bufs[(n - 1) % 3][1]), bits(frames[n - 1])), n
        t.trace_begin(W, H, n + 2, 1, bufs[(n + 1) % 3][0])
    # frames 6 and 7 are in flight: drop them; the running mean is frame 5's again
    t.trace_discard()
    img, _, _ = t.trace(W, H, 6, 1)
    assert np.array_equal(bits(img), bits(frames[5]))
    # a synchronous call drops what is in flight by itself; several samples per call work the same way
    t.trace_begin(W, H, 7, 1, bufs[0][0])
    img, _, _ = t.trace(W, H, 7, 2)
    assert np.array_equal(bits(img), bits(frames[7]))
    with pytest.raises(R.TracerError):
        t.trace_end()
    t.trace_begin(W, H, 1, 8, bufs[1][0])         # a restart (first sample 1) of eight samples in one call
    t.trace_end()
    assert np.array_equal(bits(bufs[1][1]), bits(frames[7]))
    for p, _ in bufs:
        R.Tracer.free_host(p)
    t.close()
    ref.close()


def _two_devices():
    import torch
    return [0, 1] if torch.cuda.device_count() >= 2 else [0, 0]


@pytest.mark.parametrize("devices", [None, "two"])
def test_plugin_frames_with_and_without_look_ahead(devices):
    """The headless frame loop over the HIP render mode: sample counts 1, 2, 3 ..., a camera move with a reset in between,
    a resize -- the same frames with the trace method's look-ahead on and off, on one device and over two (tile split: each
    device keeps its share of the frames in flight)."""
    devices = _two_devices() if devices == "two" else None
    product_sd_scene([WL.DRAGON_2426]).publish()
    out, rays = {}, {}
    for ahead in (False, True):
        h = R.PluginHarness(R.PluginHarness.MODE_SD, 256, 144, WL.DRAGON_2426_CAMERA, bounce_limit=4, seed=2, devices=devices)
        h.set_look_ahead(ahead)
        h.load()
        seq = []
        rays[ahead] = []
        for _ in range(4):
            h.render()
            seq.append(h.output())
            rays[ahead].append(h.last_rays())
        h.set_camera_position((0.05, 0.12, 0.40))      # RenderState::Dirty: the application resets the samples
        h.reset()
        for _ in range(3):
            h.render()
            seq.append(h.output())
        h.set_camera_position((0.0, 0.10, 0.35))       # a move WITHOUT a reset: the next sample is of the new camera
        h.render()
        seq.append(h.output())
        h.load()                                       # SceneDirty: same scene loaded again, samples go on
        h.render()
        seq.append(h.output())
        h.resize(128, 72)
        for _ in range(2):
            h.render()
            seq.append(h.output())
        assert h.sample_count() == 3
        out[ahead] = seq
        h.close()
    # the trace method's per-frame counters: a synchronous frame reports its own rays; with frames computed ahead a frame
    # reports what ran since the frame before it (up to two frames more), never the running total of the session
    per_frame = rays[False]
    assert min(per_frame) > 256 * 144 and max(per_frame) < 1.2 * min(per_frame)
    for r in rays[True]:
        assert 0 <= r <= 3.5 * max(per_frame)
    assert abs(sum(rays[True]) - sum(per_frame)) <= 2.4 * max(per_frame)
    assert len(out[True]) == len(out[False]) == 11
    for a, b in zip(out[True], out[False]):
        assert a.shape == b.shape and np.array_equal(bits(a), bits(b))
    assert not np.array_equal(out[True][3], out[True][4])
