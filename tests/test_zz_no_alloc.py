"""rt_trace_device_noalloc (the entry point for callers that cannot allocate) against rt_trace_device, through the C ABI:
same image and primary ids, bit for bit, on two calls in a row, the second with a smaller input; sizes beyond the
built-in workspace are refused."""
import numpy as np
import pytest

from helpers import R, WL, bits, product_sd_scene

pytestmark = pytest.mark.gpu


def _both(tr, torch, w, h, spp, first=1):
    dev = torch.device("cuda", 0)
    out = []
    for entry in (tr.trace_device, tr.trace_device_noalloc):
        img = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
        prim = torch.full((h, w), 0xFFFFFFFF, dtype=torch.int64, device=dev).to(torch.uint32)
        if first > 1:  # a running mean to continue from
            tr.trace_device(w, h, 1, first - 1, img.data_ptr())
        entry(w, h, first, spp, img.data_ptr(), prim.data_ptr())
        st = tr.sync()
        out.append((img.cpu().numpy(), prim.cpu().numpy(), st))
    return out


def test_noalloc_equals_trace_device_twice_second_smaller():
    import torch
    tr = R.Tracer(0)
    tr.upload(product_sd_scene([WL.DRAGON_2426]))
    tr.set_camera(WL.DRAGON_2426_CAMERA)
    tr.set_stream(torch.cuda.current_stream().cuda_stream)
    tr.set_options(bounce_limit=8, seed=5, collect_stats=True)
    for (w, h, spp, first) in ((640, 360, 3, 1), (250, 141, 5, 3)):   # the second: smaller, ragged, continues a mean
        (ai, ap, ast), (bi, bp, bst) = _both(tr, torch, w, h, spp, first)
        assert (ap != R.MISS).sum() > 1000
        assert np.array_equal(ap, bp)
        assert np.array_equal(bits(ai), bits(bi))
        assert bst.rays > 0 and bst.launches == (0 if first == 1 else 1)   # counted, not timed (the mean it continues was)
    # every traversal the library can pick runs in the same workspace; the pointer scene uses other kernels
    tr.set_options(bounce_limit=4, seed=2, traversal=R.TRAV_FAST_REFILL)
    (ai, ap, _), (bi, bp, _) = _both(tr, torch, 320, 180, 2)
    assert np.array_equal(ap, bp) and np.array_equal(bits(ai), bits(bi))
    sc = WL.build_old_scene(WL.small_sphere_scene()).build()
    tr.upload(sc)
    tr.set_camera(WL.RTIOW_CAMERA)
    tr.set_options(bounce_limit=8, estimator=R.EST_OLD_RECURSIVE, seed=3)
    (ai, ap, _), (bi, bp, _) = _both(tr, torch, 400, 225, 4)
    assert np.array_equal(ap, bp) and np.array_equal(bits(ai), bits(bi))
    # what the entry point exists for: a call leaves the device's free memory as it found it -- also the first call ever of a
    # frame size and sample count on a context that has never traced (torch.cuda.mem_get_info is hipMemGetInfo: it sees
    # every allocation of the process, whoever made it)
    fresh = R.Tracer(0)
    fresh.upload(sc)
    fresh.set_camera(WL.RTIOW_CAMERA)
    fresh.set_stream(torch.cuda.current_stream().cuda_stream)
    fresh.set_options(bounce_limit=8, estimator=R.EST_OLD_RECURSIVE, seed=3)
    for (w, h, spp) in ((512, 288, 7), (200, 100, 70), (640, 400, 1)):
        img2 = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info()
        fresh.trace_device_noalloc(w, h, 1, spp, img2.data_ptr())
        fresh.sync()
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after == free_before, (w, h, spp, free_before - free_after)
    fresh.close()
    # too large for the built-in workspace: refused, nothing launched
    img = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda:0")
    with pytest.raises(R.TracerError) as e:
        tr.trace_device_noalloc(1920, 1080, 1, 1, img.data_ptr())
    assert e.value.code == -1 and "too large" in str(e.value)
    tr.close()
    multi = R.Tracer(devices=[0, 0])   # a context over several devices has no single built-in workspace: refused
    with pytest.raises(R.TracerError):
        multi.trace_device_noalloc(64, 36, 1, 1, img.data_ptr())
    multi.close()
