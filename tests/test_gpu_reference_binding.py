"""The reference-side binding, compiled against the reference's own names (tests/mock_reference/reference_app.hpp) and
linked to the real library, renders the frames rt_trace renders for the same scene, camera and options -- from storage
vectors held at the reference's full capacity with nextIndex entries in use."""
import os
import subprocess

import numpy as np
import pytest

from helpers import R, WL, bits, product_sd_scene
from test_reference_binding_cpu import MOCK, write_scene

pytestmark = pytest.mark.gpu


def test_binding_renders_what_the_library_renders(tmp_path):
    exe = os.path.join(MOCK, "binding_check_hip")
    assert os.path.exists(exe), "make -C raytracing_amd/csrc builds it"
    ps = product_sd_scene([WL.DRAGON_2426])
    cam = WL.DRAGON_2426_CAMERA
    W, H, frames, bounce = 192, 108, 4, 5
    scene, out = tmp_path / "scene.bin", tmp_path / "out.f32"
    write_scene(scene, ps, cam, bounce)
    r = subprocess.run([exe, str(scene), str(W), str(H), str(frames), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, dtype=np.float32).reshape(H, W, 4)
    t = R.Tracer(0)
    t.upload(ps)
    t.set_camera(cam)
    t.set_options(bounce_limit=bounce, seed=1, jitter=0.001)
    want, _, _ = t.trace(W, H, 1, frames)
    t.close()
    assert np.array_equal(bits(got), bits(want))
