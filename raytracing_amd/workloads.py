"""Named workloads: scene + camera + sample settings used by the tests and by bench.py.

Scene geometry is synthetic or comes from the three OBJ files bundled with the reference
(Resources/*.obj, copied unchanged to data/models/).
"""
import os

import numpy as np

from .scene import MAT_DIELECTRIC, MAT_EMIT, MAT_LAMBERTIAN, MAT_METAL, OldScene, SdScene
from .tracer import EST_OLD_RECURSIVE, EST_SD_LOOP, Camera

MODELS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "models")
DRAGON_2426 = os.path.join(MODELS, "TheStanfordDragon2426.obj")
DRAGON_18520 = os.path.join(MODELS, "TheStanfordDragon18520.obj")
CYLINDER = os.path.join(MODELS, "SphereSingle.obj")  # a 32-segment cylinder despite its name

# cameras that frame the two dragons (the reference default camera, Renderer/Renderer.hpp:61, is DEFAULT_CAMERA)
DEFAULT_CAMERA = Camera(1.0, (0.0, 0.0, 1.0), 2.0, 16.0 / 9.0, (0.0, 0.0, 0.0))
DRAGON_2426_CAMERA = Camera(1.0, (0.0, 0.10, 0.35), 2.0, 16.0 / 9.0, (-0.006, 0.073, -0.014))
DRAGON_18520_CAMERA = Camera(1.0, (0.0, 0.10, -0.55), 2.0, 16.0 / 9.0, (-0.006, 0.072, -0.915))
RTIOW_CAMERA = Camera(0.5625 / np.tan(np.radians(10.0)), (13.0, 2.0, 3.0), 2.0, 16.0 / 9.0, (0.0, 0.0, 0.0))


def _pcg(v):
    v = np.uint64(v) & np.uint64(0xFFFFFFFF)
    s = (v * np.uint64(747796405) + np.uint64(2891336453)) & np.uint64(0xFFFFFFFF)
    w = (((s >> ((s >> np.uint64(28)) + np.uint64(4))) ^ s) * np.uint64(277803737)) & np.uint64(0xFFFFFFFF)
    return int(((w >> np.uint64(22)) ^ w) & np.uint64(0xFFFFFFFF))


class _Rand:
    """Deterministic uniform stream for scene generation (same hash as the tracer's counter RNG)."""

    def __init__(self, seed):
        self.key, self.n = _pcg(seed), 0

    def __call__(self):
        self.n += 1
        return np.float32((_pcg(self.key + self.n * 0x9E3779B9) >> 8) / 16777216.0)


def rtiow_spheres(seed=42, grid=11):
    """The 'Ray Tracing in One Weekend' final scene: ground sphere, (2*grid)^2 jittered small spheres
    (80 % Lambertian, 15 % metal, 5 % glass), three large spheres.  Returns a list of
    (center, radius, kind, albedo4, param)."""
    r = _Rand(seed)
    out = [((0.0, -1000.0, 0.0), 1000.0, MAT_LAMBERTIAN, (0.5, 0.5, 0.5, 1.0), 0.0)]
    for a in range(-grid, grid):
        for b in range(-grid, grid):
            choose = r()
            c = (np.float32(a) + np.float32(0.9) * r(), np.float32(0.2), np.float32(b) + np.float32(0.9) * r())
            if np.sqrt((c[0] - 4.0) ** 2 + (c[1] - 0.2) ** 2 + c[2] ** 2) <= 0.9:
                continue
            c = tuple(float(x) for x in c)
            if choose < 0.8:
                alb = (float(r() * r()), float(r() * r()), float(r() * r()), 1.0)
                out.append((c, 0.2, MAT_LAMBERTIAN, alb, 0.0))
            elif choose < 0.95:
                alb = tuple(float(np.float32(0.5) + np.float32(0.5) * r()) for _ in range(3)) + (1.0,)
                out.append((c, 0.2, MAT_METAL, alb, float(np.float32(0.5) * r())))
            else:
                out.append((c, 0.2, MAT_DIELECTRIC, (1.0, 1.0, 1.0, 1.0), 1.5))
    out.append(((0.0, 1.0, 0.0), 1.0, MAT_DIELECTRIC, (1.0, 1.0, 1.0, 1.0), 1.5))
    out.append(((-4.0, 1.0, 0.0), 1.0, MAT_LAMBERTIAN, (0.4, 0.2, 0.1, 1.0), 0.0))
    out.append(((4.0, 1.0, 0.0), 1.0, MAT_METAL, (0.7, 0.6, 0.5, 1.0), 0.0))
    return out


def small_sphere_scene():
    """Five spheres: ground, glass, Lambertian, metal, emitter."""
    return [((0.0, -1000.0, 0.0), 1000.0, MAT_LAMBERTIAN, (0.5, 0.5, 0.5, 1.0), 0.0),
            ((0.0, 1.0, 0.0), 1.0, MAT_DIELECTRIC, (1.0, 1.0, 1.0, 1.0), 1.5),
            ((-4.0, 1.0, 0.0), 1.0, MAT_LAMBERTIAN, (0.4, 0.2, 0.1, 1.0), 0.0),
            ((4.0, 1.0, 0.0), 1.0, MAT_METAL, (0.7, 0.6, 0.5, 1.0), 0.0),
            ((0.0, 3.5, 2.0), 0.5, MAT_EMIT, (4.0, 4.0, 4.0, 1.0), 0.0)]


def build_old_scene(spheres, scene=None):
    s = scene if scene is not None else OldScene()
    for c, rad, kind, alb, param in spheres:
        s.add_sphere(c, rad, kind, alb, param)
    return s


def dragon_grid_tris(tris24, k, spacing=0.25):
    """k x k translated copies of a triangle soup, flattened into one list (copy (i, j) moved by
    (i*spacing, 0, j*spacing))."""
    t = np.ascontiguousarray(tris24, dtype=np.float32).reshape(-1, 3, 8)
    out = np.empty((k * k,) + t.shape, dtype=np.float32)
    for i in range(k):
        for j in range(k):
            c = t.copy()
            c[:, :, 0] += np.float32(i * spacing)
            c[:, :, 2] += np.float32(j * spacing)
            out[i * k + j] = c
    return out.reshape(-1, 24)


class Workload:
    def __init__(self, name, scene, camera, estimator, width, height, spp, bounce_limit, seed=1, jitter=0.001,
                 obj_path=None, field_k=0):
        self.name, self.scene, self.camera, self.estimator = name, scene, camera, estimator
        self.width, self.height, self.spp, self.bounce_limit = width, height, spp, bounce_limit
        self.seed, self.jitter = seed, jitter
        # how a checker rebuilds the same scene: the OBJ file, and the side of the grid of copies (0 = the file as is)
        self.obj_path, self.field_k = obj_path, field_k


def field_camera(k, spacing=0.25):
    """Looks down at the k x k dragon field from in front of it."""
    half = np.float32(spacing * (k - 1) / 2)
    c = (float(half) - 0.006, 0.07, -0.915 + float(half))
    s = 0.9 * spacing * k
    return Camera(2.0, (c[0], c[1] + 0.55 * s, c[2] + 1.0 * s), 2.0, 16.0 / 9.0, c)


def _load_obj_tris(path):
    """Triangle soup (n, 24) of an OBJ file through the product's own reader (csrc/host/obj_reader.cpp)."""
    sc = SdScene().load_obj(path).finalize()
    t = sc.tris()
    out = np.zeros((len(t), 3, 8), dtype=np.float32)
    out[:, :, 0:3] = t["pos"]
    out[:, :, 3:6] = t["nrm"]
    out[:, :, 6:8] = t["uv"]
    return out.reshape(-1, 24)


def dragon_field(name, k, width, height, spp):
    """k x k copies of the 18520-triangle dragon flattened into one triangle list (k = 8: 1.19 M triangles,
    k = 23: 9.8 M), one flat-array scene under the reference's median-split BVH."""
    tris = dragon_grid_tris(_load_obj_tris(DRAGON_18520), k)
    sc = SdScene().add_model(tris).finalize()
    return Workload(name, sc, field_camera(k), EST_SD_LOOP, width, height, spp, 8, obj_path=DRAGON_18520, field_k=k)


def make_workload(name):
    """Workloads named after the configurations they realise."""
    if name == "dragon18520_1080p":  # the headline: 18520-triangle Stanford Dragon, 1080p
        sc = SdScene().load_obj(DRAGON_18520).finalize()
        return Workload(name, sc, DRAGON_18520_CAMERA, EST_SD_LOOP, 1920, 1080, 16, 8, obj_path=DRAGON_18520)
    if name == "dragon2426_1080p":
        sc = SdScene().load_obj(DRAGON_2426).finalize()
        return Workload(name, sc, DRAGON_2426_CAMERA, EST_SD_LOOP, 1920, 1080, 16, 8, obj_path=DRAGON_2426)
    if name == "dragon18520_4k":
        sc = SdScene().load_obj(DRAGON_18520).finalize()
        return Workload(name, sc, DRAGON_18520_CAMERA, EST_SD_LOOP, 3840, 2160, 256, 8, obj_path=DRAGON_18520)
    if name == "dragon_field_k8_1080p":   # 1.19 M triangles
        return dragon_field(name, 8, 1920, 1080, 16)
    if name == "dragon_field_k8_4k":
        return dragon_field(name, 8, 3840, 2160, 1024)
    if name == "dragon_field_k23_4k":     # 9.8 M triangles: node and triangle records no longer fit the L2
        return dragon_field(name, 23, 3840, 2160, 1024)
    if name == "dragon_field_k23_1080p":
        return dragon_field(name, 23, 1920, 1080, 16)
    if name == "rtiow_1080p":
        sc = build_old_scene(rtiow_spheres()).build()
        return Workload(name, sc, RTIOW_CAMERA, EST_OLD_RECURSIVE, 1920, 1080, 64, 8, seed=42)
    if name == "plumbing_400x225":
        sc = build_old_scene(small_sphere_scene())
        sc.add_obj(CYLINDER, MAT_LAMBERTIAN, (0.8, 0.3, 0.3, 1.0), 0.0)  # material of ModelLoader.hpp:62
        sc.build()
        return Workload(name, sc, Camera(3.19, (13.0, 2.0, 3.0), 2.0, 16.0 / 9.0, (0.0, 0.5, 0.0)), EST_OLD_RECURSIVE,
                        400, 225, 8, 8)
    raise KeyError(name)
