"""The reference-side binding (raytracing_amd/csrc/host/reference_binding.hpp) compiles against a header that declares the
reference's own names (tests/mock_reference/reference_app.hpp: Storage::SdScene, nodeStorage.nodes / nextIndex, glm-style
camera members, RenderPipeline ...) and hands the C ABI the right arrays, counts and camera: checked with a recording
stand-in of the ABI, no GPU."""
import os
import subprocess

import numpy as np

from helpers import R, WL, product_sd_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_reference")


def write_scene(path, ps, cam, bounce):
    nodes, tris = ps.nodes(), ps.tris()
    with open(path, "wb") as fh:
        fh.write(np.array([len(nodes), ps.root, len(tris)], dtype="<u4").tobytes())
        fh.write(nodes.tobytes())
        fh.write(tris.tobytes())
        fh.write(np.array([cam.focal_length, *cam.position, *cam.look_at, cam.width, cam.aspect_ratio], dtype="<f4").tobytes())
        fh.write(np.array([bounce], dtype="<i4").tobytes())


def field_hash(records, used):
    """recording_rt.cpp's hash: FNV-1a over the first `used` bytes of each record, chained by * 31."""
    raw = np.frombuffer(records.tobytes(), dtype=np.uint8).reshape(len(records), -1)[:, :used].astype(np.uint64)
    mask = (1 << 64) - 1
    h = np.full(len(records), 1469598103934665603, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for k in range(used):
            h = (h ^ raw[:, k]) * np.uint64(1099511628211)
    total = 1469598103934665603
    for v in h.tolist():
        total = (total * 31 + v) & mask
    return total


def test_binding_compiles_against_the_reference_names():
    # the committed sources, not a binary that may be older than them
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I", MOCK,
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "raytracing_amd", "csrc", "host"),
                        os.path.join(MOCK, "binding_check.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # and nothing of this repository's stand-ins is named by the binding
    text = open(os.path.join(ROOT, "raytracing_amd", "csrc", "host", "reference_binding.hpp")).read()
    for name in ("render_plugin.hpp", "HeadlessApp", "SdScene_", "sd_scene.hpp"):
        assert name not in text


def test_binding_hands_over_the_entries_in_use(tmp_path):
    exe = os.path.join(MOCK, "binding_check_rec")
    assert os.path.exists(exe), "make -C raytracing_amd/csrc builds it"
    ps = product_sd_scene([WL.DRAGON_2426])
    cam = WL.DRAGON_2426_CAMERA
    scene, log, out = tmp_path / "scene.bin", tmp_path / "calls.txt", tmp_path / "out.f32"
    write_scene(scene, ps, cam, 5)
    r = subprocess.run([exe, str(scene), "64", "36", "3", str(out)], env=dict(os.environ, RT_RECORD=str(log)),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    calls = [ln.split(" ", 1) for ln in open(log).read().splitlines()]
    kinds = [c[0] for c in calls]
    rec = {c[0]: dict(kv.split("=", 1) for kv in c[1].split()) for c in calls if len(c) > 1}
    up = rec["upload"]
    # the storages hold 2^21 - 1 nodes and 2^20 triangles; what goes over is [0, nextIndex)
    assert int(up["n_nodes"]) == len(ps.nodes()) and int(up["n_tris"]) == len(ps.tris()) and int(up["root"]) == ps.root
    assert int(up["nodes_hash"]) == field_hash(ps.nodes(), 33) and int(up["tris_hash"]) == field_hash(ps.tris(), 98)
    assert up["order"] == "0" and up["spheres"] == "0" and up["materials"] == "0"   # RT_ORDER_RIGHT_FIRST, no spheres
    c = rec["camera"]
    assert np.float32(c["focal"]) == np.float32(cam.focal_length)
    assert [np.float32(v) for v in c["pos"].split(",")] == [np.float32(v) for v in cam.position]
    assert [np.float32(v) for v in c["look"].split(",")] == [np.float32(v) for v in cam.look_at]
    assert np.float32(c["width"]) == np.float32(cam.width) and np.float32(c["height"]) == np.float32(cam.height)
    o = rec["options"]
    assert o["bounce_limit"] == "5" and o["estimator"] == "0" and o["seed"] == "1" and o["tile_world"] == "1"
    assert np.float32(o["jitter"]) == np.float32(0.001)           # Renderer/TraceMethods.cpp:48
    # the frame loop: sample 1 is begun and handed over, 2 and 3 are begun ahead; every frame the caller sees is its own
    assert kinds.index("upload") < kinds.index("trace_begin")
    begun = [int(dict(kv.split("=") for kv in c[1].split())["first"]) for c in calls if c[0] == "trace_begin"]
    ended = [int(c[1].split("=")[1]) for c in calls if c[0] == "trace_end"]
    assert ended == [1, 2, 3] and begun[:3] == [1, 2, 3] and kinds[-1] == "destroy"
    img = np.fromfile(out, dtype=np.float32)
    assert img.size == 64 * 36 * 4 and np.all(img == 3.0)
