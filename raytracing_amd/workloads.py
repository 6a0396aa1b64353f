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
# configuration 4: the 18 520-triangle dragon with the full-resolution stand-in 0.22 to its right, both in frame
DRAGON_PAIR_CAMERA = Camera(1.0, (0.105, 0.13, -0.52), 2.0, 16.0 / 9.0, (0.105, 0.06, -0.915))
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
                 models=None, spheres=None, bench_spp=0, description=""):
        self.name, self.scene, self.camera, self.estimator = name, scene, camera, estimator
        self.width, self.height, self.spp, self.bounce_limit = width, height, spp, bounce_limit
        self.seed, self.jitter = seed, jitter
        # samples per pixel of one bench.py step (a step should last tens of milliseconds); default: the configuration's spp
        self.bench_spp = bench_spp or spp
        self.description = description
        # how a checker rebuilds the same scene elsewhere: the triangle soups (n, 24) of the flat scene's models in order, each
        # one mesh, or the sphere list of the pointer scene
        self.models, self.spheres = models, spheres

    def scene_bytes(self):
        """What the walk touches, as uploaded.  Flat scenes (the loop estimator): 16 B per grid record of a full binary
        tree over the triangles, 64 B per leaf record (exact box + triangle).  Pointer scenes: 32 B per fast-tree
        record, 48 B of positions per triangle."""
        prims = len(self.scene.tris()) + len(self.scene.spheres())
        if self.estimator == EST_SD_LOOP:
            return 16 * max(2 * prims - 1, 1) + 64 * len(self.scene.tris())
        return 32 * max(2 * prims - 1, 1) + 48 * len(self.scene.tris())


def subdivide(tris24, levels=2, centroid_split=True):
    """Refines a triangle soup (n, 24): `levels` times every triangle is split in four at its edge midpoints, then
    (centroid_split) in three at its centroid.  Positions and uvs are interpolated linearly in fp32, normals are
    interpolated and re-normalised; the surface stays the same set of points, covered by 4**levels * 3 times as many
    triangles."""
    t = np.ascontiguousarray(tris24, dtype=np.float32).reshape(-1, 3, 8)

    def mid(a, b):
        m = (a + b) * np.float32(0.5)
        n = m[:, 3:6]
        ln = np.sqrt((n * n).sum(axis=1, keepdims=True))
        m[:, 3:6] = np.where(ln > 0, n / np.maximum(ln, np.float32(1e-30)), a[:, 3:6])
        return m.astype(np.float32)

    for _ in range(levels):
        a, b, c = t[:, 0], t[:, 1], t[:, 2]
        ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
        t = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1),
                      np.stack([ab, bc, ca], 1)], 1).reshape(-1, 3, 8)
    if centroid_split:
        a, b, c = t[:, 0], t[:, 1], t[:, 2]
        g = ((a + b + c) * np.float32(1.0 / 3.0)).astype(np.float32)
        n = g[:, 3:6]
        ln = np.sqrt((n * n).sum(axis=1, keepdims=True))
        g[:, 3:6] = np.where(ln > 0, n / np.maximum(ln, np.float32(1e-30)), a[:, 3:6])
        t = np.stack([np.stack([a, b, g], 1), np.stack([b, c, g], 1), np.stack([c, a, g], 1)], 1).reshape(-1, 3, 8)
    return np.ascontiguousarray(t.reshape(-1, 24), dtype=np.float32)


def ground_quad(y=0.0, x0=-1.0, x1=1.3, z0=-2.2, z1=0.4):
    """Two triangles in the plane y = const, normal +y."""
    def v(x, z):
        return [x, y, z, 0.0, 1.0, 0.0, 0.0, 0.0]
    return np.array([v(x0, z0) + v(x0, z1) + v(x1, z1), v(x0, z0) + v(x1, z1) + v(x1, z0)], dtype=np.float32)


FULL_DRAGON_SUBSTITUTE = ("substitute for the reference's missing TheStanfordDragon.obj (871 414 triangles): the 18 520-triangle "
                          "mesh with every triangle split in four twice and in three once (888 960 triangles), moved 0.22 in x")


def dragon_pair_models():
    """Configuration 4's scene as three models: the 18 520-triangle dragon, a stand-in for the full-resolution dragon
    beside it, and a ground quad under both (the dragon's lowest vertex is at y = 0.0001)."""
    base = _load_obj_tris(DRAGON_18520)
    full = subdivide(base, 2, True)
    full = full.reshape(-1, 3, 8).copy()
    full[:, :, 0] += np.float32(0.22)
    return [base, np.ascontiguousarray(full.reshape(-1, 24)), ground_quad()]


def sd_workload(name, models, camera, width, height, spp, **kw):
    sc = SdScene()
    for m in models:
        sc.add_model(m)
    sc.finalize()
    return Workload(name, sc, camera, EST_SD_LOOP, width, height, spp, 8, models=models, **kw)


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


def dragon_field(name, k, width, height, spp, **kw):
    """k x k copies of the 18520-triangle dragon flattened into one triangle list (k = 8: 1.19 M triangles,
    k = 23: 9.8 M), one flat-array scene under the reference's median-split BVH."""
    tris = dragon_grid_tris(_load_obj_tris(DRAGON_18520), k)
    return sd_workload(name, [tris], field_camera(k), width, height, spp,
                       description=f"{k} x {k} translated copies of TheStanfordDragon18520.obj flattened into one triangle list", **kw)


def make_workload(name):
    """Workloads named after the configurations they realise."""
    if name == "dragon18520_1080p":  # the headline: 18520-triangle Stanford Dragon, 1080p
        return sd_workload(name, [_load_obj_tris(DRAGON_18520)], DRAGON_18520_CAMERA, 1920, 1080, 16, bench_spp=512,
                           description="TheStanfordDragon18520.obj alone (no ground: about 92 % of the camera rays leave to the sky)")
    if name == "dragon2426_1080p":
        return sd_workload(name, [_load_obj_tris(DRAGON_2426)], DRAGON_2426_CAMERA, 1920, 1080, 16, bench_spp=512,
                           description="TheStanfordDragon2426.obj alone")
    if name in ("dragon18520_4k", "dragon18520_full_ground_1080p"):
        # configuration 4 as stated: Dragon18520 + the full dragon on a ground quad, 4K, 256 spp
        w, h = (3840, 2160) if name.endswith("4k") else (1920, 1080)
        return sd_workload(name, dragon_pair_models(), DRAGON_PAIR_CAMERA, w, h, 256 if name.endswith("4k") else 16,
                           bench_spp=256 if name.endswith("4k") else 64,
                           description="TheStanfordDragon18520.obj + " + FULL_DRAGON_SUBSTITUTE + ", on a ground quad; every ray hits or bounces")
    if name == "dragon18520_ground_1080p":
        return sd_workload(name, [_load_obj_tris(DRAGON_18520), ground_quad()], DRAGON_PAIR_CAMERA, 1920, 1080, 16, bench_spp=64,
                           description="TheStanfordDragon18520.obj on configuration 4's ground quad (a scene within the L2s in which every ray hits or bounces)")
    if name == "dragon18520_bare_4k":
        return sd_workload(name, [_load_obj_tris(DRAGON_18520)], DRAGON_18520_CAMERA, 3840, 2160, 256,
                           description="TheStanfordDragon18520.obj alone at 4K")
    if name == "dragon_field_k8_1080p":   # 1.19 M triangles
        return dragon_field(name, 8, 1920, 1080, 16, bench_spp=64)
    if name == "dragon_field_k8_4k":
        return dragon_field(name, 8, 3840, 2160, 1024)
    if name == "dragon_field_k23_4k":     # 9.8 M triangles: node and triangle records no longer fit the L2
        return dragon_field(name, 23, 3840, 2160, 1024)
    if name == "dragon_field_k23_1080p":
        return dragon_field(name, 23, 1920, 1080, 16, bench_spp=64)
    if name == "rtiow_1080p":
        sph = rtiow_spheres()
        sc = build_old_scene(sph).build()
        return Workload(name, sc, RTIOW_CAMERA, EST_OLD_RECURSIVE, 1920, 1080, 64, 8, seed=42, spheres=sph, bench_spp=128,
                        description="'Ray Tracing in One Weekend' final scene, 485 spheres, Lambertian / metal / dielectric")
    if name == "cylinder_100x60":   # the smallest flat scene there is: what the CPU tests of the multi-rank path render
        return sd_workload(name, [_load_obj_tris(CYLINDER)], DEFAULT_CAMERA, 100, 60, 2, bench_spp=2,
                           description="SphereSingle.obj (a 124-triangle cylinder) from the reference's default camera")
    if name == "plumbing_400x225":
        sph = small_sphere_scene()
        sc = build_old_scene(sph)
        sc.add_obj(CYLINDER, MAT_LAMBERTIAN, (0.8, 0.3, 0.3, 1.0), 0.0)  # material of ModelLoader.hpp:62
        sc.build()
        return Workload(name, sc, Camera(3.19, (13.0, 2.0, 3.0), 2.0, 16.0 / 9.0, (0.0, 0.5, 0.0)), EST_OLD_RECURSIVE,
                        400, 225, 8, 8, spheres=sph, description="five spheres + SphereSingle.obj (a cylinder)")
    raise KeyError(name)
