"""The host-side native code under the compiler's sanitizers (host code only; no GPU involved).

* AddressSanitizer + UndefinedBehaviorSanitizer: OBJ reader, both scene builders (copy / assignment included), flat
  packers and the fast-tree builder, driven through their normal call sequence and their error paths.
* ThreadSanitizer: the fast-tree builder's shared-thread build.
The sanitizer runtimes are linked statically so that nothing depends on the order of preloaded libraries."""
import os
import subprocess

import pytest

from helpers import ROOT, WL

CSRC = os.path.join(ROOT, "raytracing_amd", "csrc")
NATIVE = os.path.join(ROOT, "tests", "native")


def _build(tmp_path, name, flags, sources):
    exe = str(tmp_path / name)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-pthread", "-fno-sanitize-recover=all", *flags,
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(CSRC, "host"), *sources, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("cannot find" in r.stderr or "unrecognized" in r.stderr):
        pytest.skip("sanitizer runtime not available: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr
    return exe


def test_host_builders_under_asan_and_ubsan(tmp_path):
    exe = _build(tmp_path, "host_sanitized", ["-fsanitize=address,undefined", "-static-libasan"],
                 [os.path.join(NATIVE, "host_sanitized.cpp")] + [os.path.join(CSRC, "host", f) for f in ("obj_reader.cpp", "old_scene.cpp", "sd_scene.cpp")])
    for models in ([WL.DRAGON_2426], [WL.CYLINDER, WL.DRAGON_2426]):
        out = subprocess.run([exe, *models], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr[-2000:]


def test_fast_tree_shared_build_under_tsan(tmp_path):
    exe = _build(tmp_path, "fast_bvh_tsan", ["-fsanitize=thread", "-static-libtsan"], [os.path.join(NATIVE, "fast_bvh_check.cpp")])
    out = subprocess.run([exe, "small"], capture_output=True, text=True, timeout=600)
    if out.returncode != 0 and "unexpected memory mapping" in out.stderr:
        pytest.skip("the thread sanitizer cannot run in this address-space layout")
    assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr and out.stdout.count("ok ") == 6, out.stdout + out.stderr[-2000:]
