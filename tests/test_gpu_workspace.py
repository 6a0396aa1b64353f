"""What a trace call allocates: the wavefront workspace of the default flow on a flat scene is the colour slots, the camera
stage's 16-byte records and the `deferred` words -- no queue of path records, because the re-fill kernel follows every path to
its end -- and rt_reserve's worst case (any scene, any options) stays what it was."""
import pytest

from helpers import R, WL, product_sd_scene

pytestmark = pytest.mark.gpu
GB = 1e9


def test_workspace_of_a_1080p_64_sample_call():
    import torch
    W, H, SPP = 1920, 1080, 64
    tr = R.Tracer(0)
    tr.upload(product_sd_scene([WL.DRAGON_18520]))
    tr.set_camera(WL.DRAGON_18520_CAMERA)
    tr.set_options(bounce_limit=8, seed=1)
    tr.trace(64, 64, 1, 1, want_image=False)            # context-level buffers exist
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    tr.trace(W, H, 1, SPP, want_image=False)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    used = free0 - free1
    items = (W // 8) * ((H + 7) // 8) * 64 * SPP
    # colours 16 B + camera record 16 B per pixel sample (+ queue slack, deferred words, the image): 4.3 GB
    assert items * 32 <= used <= 8 * GB, used / GB
    # a second call of the same size allocates nothing
    tr.trace(W, H, 1, SPP, want_image=False)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free1
    # several chunks per call alternate between two workspaces: twice that, and no more afterwards
    tr.trace(W, H, 1, 3 * SPP, want_image=False)
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert free0 - free2 <= 2 * used + 64e6 and free0 - free2 <= 9 * GB
    tr.trace(W, H, 1, 3 * SPP, want_image=False)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free2
    tr.close()
