"""ctypes front end of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the cpu_baseline leg
of bench.py as the checker.  The product package never imports this module.  PARITY UNPINNED: see
oracle/rt_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")

TRIG_LIBM, TRIG_POLY = 0, 1
TRAV_LOOP, TRAV_RECURSIVE, TRAV_BRUTE, TRAV_LEAVES, TRAV_LIMITED = 0, 1, 2, 3, 4
ACCUM_RECURSIVE, ACCUM_FORWARD = 0, 1
MAT_LAMBERTIAN, MAT_METAL, MAT_EMIT, MAT_DIELECTRIC = 0, 1, 2, 4
MISS = 0xFFFFFFFF

NODE_DTYPE = np.dtype([("left", "<u4"), ("right", "<u4"), ("pmin", "<f4", 3), ("pmax", "<f4", 3), ("flags", "<u4")])
TRI_DTYPE = np.dtype([("pos", "<f4", (3, 3)), ("nrm", "<f4", (3, 3)), ("uv", "<f4", (3, 2)), ("mat", "<u4")])
assert NODE_DTYPE.itemsize == 36 and TRI_DTYPE.itemsize == 100


class Camera(C.Structure):
    _fields_ = [("focal_length", C.c_float), ("position", C.c_float * 3), ("look_at", C.c_float * 3),
                ("width", C.c_float), ("height", C.c_float)]


class Options(C.Structure):
    _fields_ = [("bounce_limit", C.c_int32), ("seed", C.c_uint32), ("jitter", C.c_float), ("trig", C.c_int32),
                ("traversal", C.c_int32), ("threads", C.c_int32), ("lock_per_pixel", C.c_int32),
                ("accum", C.c_int32), ("tile_rank", C.c_int32), ("tile_world", C.c_int32), ("sample_offset", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("sph_tests", C.c_uint64), ("tri_hits", C.c_uint64), ("seconds", C.c_double)]


class Hit(C.Structure):
    _fields_ = [("hit", C.c_int32), ("t", C.c_float), ("pos", C.c_float * 3), ("normal", C.c_float * 3),
                ("prim", C.c_uint32), ("mat", C.c_uint32), ("box_tests", C.c_uint32), ("tri_tests", C.c_uint32)]


def build(force=False):
    """Compile the oracle with its Makefile (g++ only)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rt_oracle.cpp")):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, fp, up = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.ora_sd_create.restype = vp
        L.ora_sd_destroy.argtypes = [vp]
        L.ora_sd_add_model.argtypes = [vp, fp, up, C.c_uint32]
        L.ora_sd_finalize.argtypes = [vp]
        L.ora_sd_counts.argtypes = [vp, up, up, up]
        L.ora_sd_get_nodes.argtypes = [vp, vp]
        L.ora_sd_get_tris.argtypes = [vp, vp]
        L.ora_sd_flat_nodes.argtypes = [vp, fp]
        L.ora_sd_flat_tris.argtypes = [vp, fp]
        L.ora_flat_node_unpack.argtypes = [fp, C.c_uint32, vp]
        L.ora_flat_tri_unpack.argtypes = [fp, C.c_uint32, vp]
        L.ora_sd_intersect.argtypes = [vp, fp, fp, C.c_int, C.POINTER(Hit)]
        L.ora_sd_set_limit.argtypes = [vp, fp, C.POINTER(C.c_uint8), C.c_uint32]
        L.ora_take_limit_counts.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ora_take_limit_counts.restype = None
        L.ora_sd_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Options), C.c_int, C.c_int, C.c_int, C.c_int,
                                    fp, up, C.POINTER(Stats)]
        L.ora_old_create.restype = vp
        L.ora_old_destroy.argtypes = [vp]
        L.ora_old_add_sphere.argtypes = [vp, fp, C.c_float, C.c_uint32, fp, C.c_float]
        L.ora_old_add_mesh.argtypes = [vp, fp, C.c_uint32, C.c_uint32, fp, C.c_float]
        L.ora_old_build.argtypes = [vp]
        L.ora_old_render.argtypes = L.ora_sd_render.argtypes
        L.ora_old_intersect.argtypes = [vp, fp, fp, C.c_int, C.POINTER(Hit)]
        L.ora_old_flatten.argtypes = [vp, fp]
        L.ora_old_flatten.restype = C.c_uint32
        L.ora_box_test.argtypes = [fp, fp, fp, fp, C.c_float, C.c_float]
        L.ora_rng_u32.argtypes = [C.c_uint32] * 4
        L.ora_rng_u32.restype = C.c_uint32
        L.ora_sincos.argtypes = [C.c_float, C.c_int, fp, fp]
        L.ora_camera_dir.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, fp]
        L.ora_cosine_dir.argtypes = [fp, C.c_float, C.c_float, C.c_int, fp]
        L.ora_vec_op.argtypes = [C.c_int, fp, fp, fp, fp, fp]
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def make_camera(focal, position, look_at, width, aspect):
    cam = Camera()
    cam.focal_length = focal
    cam.position[:] = [float(x) for x in position]
    cam.look_at[:] = [float(x) for x in look_at]
    cam.width = width
    cam.height = np.float32(width) / np.float32(aspect)  # Include/Camera.hpp:26
    return cam


def make_options(bounce_limit=4, seed=1, jitter=0.001, trig=TRIG_POLY, traversal=TRAV_LOOP, threads=8,
                 lock_per_pixel=0, accum=ACCUM_RECURSIVE, tile_rank=0, tile_world=1, sample_offset=0):
    return Options(bounce_limit, seed, jitter, trig, traversal, threads, lock_per_pixel, accum, tile_rank,
                   tile_world, sample_offset)


def _render(fn, handle, cam, opt, width, height, first_sample, n_samples, image, want_primary):
    if image is None:
        image = np.zeros((height, width, 4), dtype=np.float32)
    assert image.dtype == np.float32 and image.flags["C_CONTIGUOUS"] and image.size == width * height * 4
    prim = np.full((height, width), MISS, dtype=np.uint32) if want_primary else None
    stats = Stats()
    rc = fn(handle, C.byref(cam), C.byref(opt), width, height, first_sample, n_samples,
            image.ctypes.data_as(C.POINTER(C.c_float)),
            prim.ctypes.data_as(C.POINTER(C.c_uint32)) if want_primary else None, C.byref(stats))
    if rc != 0:
        raise RuntimeError("oracle render failed (scene not loaded or bad arguments)")
    return image, prim, stats


class SdScene:
    """Flat-array scene of CPURayTrace/SimplifiedData.{hpp,cpp}."""

    def __init__(self):
        self.h = lib().ora_sd_create()

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_sd_destroy(self.h)
            self.h = None

    def add_model(self, tris24, mesh_sizes=None):
        tris24 = np.ascontiguousarray(tris24, dtype=np.float32).reshape(-1, 24)
        if mesh_sizes is None:
            mesh_sizes = [len(tris24)]
        ms = np.ascontiguousarray(mesh_sizes, dtype=np.uint32)
        assert int(ms.sum()) == len(tris24)
        rc = lib().ora_sd_add_model(self.h, tris24.ctypes.data_as(C.POINTER(C.c_float)),
                                    ms.ctypes.data_as(C.POINTER(C.c_uint32)), len(ms))
        if rc != 0:
            raise RuntimeError("oracle: Build Failed. end - start <= 0")

    def finalize(self):
        if lib().ora_sd_finalize(self.h) != 0:
            raise RuntimeError("oracle: Build Failed. end - start <= 0")

    def counts(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().ora_sd_counts(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def nodes(self):
        n, _, _ = self.counts()
        out = np.zeros(n, dtype=NODE_DTYPE)
        lib().ora_sd_get_nodes(self.h, out.ctypes.data)
        return out

    def tris(self):
        _, n, _ = self.counts()
        out = np.zeros(n, dtype=TRI_DTYPE)
        lib().ora_sd_get_tris(self.h, out.ctypes.data)
        return out

    def flat(self):
        n, t, _ = self.counts()
        fn = np.zeros((n, 9), dtype=np.float32)
        ft = np.zeros((t, 25), dtype=np.float32)
        lib().ora_sd_flat_nodes(self.h, fn.ctypes.data_as(C.POINTER(C.c_float)))
        lib().ora_sd_flat_tris(self.h, ft.ctypes.data_as(C.POINTER(C.c_float)))
        return fn, ft

    def intersect(self, origin, direction, traversal=TRAV_LOOP):
        o, op = _f(origin)
        d, dp = _f(direction)
        h = Hit()
        lib().ora_sd_intersect(self.h, op, dp, traversal, C.byref(h))
        return h

    def set_limit(self, constants9, class_by_triangle):
        """Distance limit of TRAV_LIMITED: c1[3], 2 c2[3], (c2 W + c3)[3] and the size class of every triangle."""
        c = np.ascontiguousarray(constants9, dtype=np.float32).reshape(9)
        k = np.ascontiguousarray(class_by_triangle, dtype=np.uint8)
        rc = lib().ora_sd_set_limit(self.h, c.ctypes.data_as(C.POINTER(C.c_float)), k.ctypes.data_as(C.POINTER(C.c_uint8)), len(k))
        if rc != 0:
            raise ValueError("one class per triangle of the scene is needed")

    def render(self, cam, opt, width, height, first_sample=1, n_samples=1, image=None, want_primary=False):
        return _render(lib().ora_sd_render, self.h, cam, opt, width, height, first_sample, n_samples, image,
                       want_primary)


class OldScene:
    """Pointer-based scene of Objects.hpp / Mesh.hpp / BVH.hpp with virtual materials."""

    def __init__(self):
        self.h = lib().ora_old_create()

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_old_destroy(self.h)
            self.h = None

    def add_sphere(self, center, radius, kind, albedo=(1, 1, 1, 1), param=0.0):
        c, cp = _f(center)
        a, ap = _f(albedo)
        lib().ora_old_add_sphere(self.h, cp, radius, kind, ap, param)

    def add_mesh(self, tris24, kind, albedo=(1, 1, 1, 1), param=0.0):
        t = np.ascontiguousarray(tris24, dtype=np.float32).reshape(-1, 24)
        a, ap = _f(albedo)
        if lib().ora_old_add_mesh(self.h, t.ctypes.data_as(C.POINTER(C.c_float)), len(t), kind, ap, param) != 0:
            raise RuntimeError("oracle: empty mesh")

    def build(self):
        lib().ora_old_build(self.h)

    def intersect(self, origin, direction, traversal=TRAV_LOOP):
        o, op = _f(origin)
        d, dp = _f(direction)
        h = Hit()
        lib().ora_old_intersect(self.h, op, dp, traversal, C.byref(h))
        return h

    def flatten(self):
        n = lib().ora_old_flatten(self.h, None)
        out = np.zeros((n, 8), dtype=np.float32)
        lib().ora_old_flatten(self.h, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def render(self, cam, opt, width, height, first_sample=1, n_samples=1, image=None, want_primary=False):
        return _render(lib().ora_old_render, self.h, cam, opt, width, height, first_sample, n_samples, image,
                       want_primary)


def unpack_flat_node(flat_nodes, index):
    out = np.zeros(1, dtype=NODE_DTYPE)
    fa, fp = _f(flat_nodes)
    lib().ora_flat_node_unpack(fp, index, out.ctypes.data)
    return out[0]


def unpack_flat_tri(flat_tris, index):
    out = np.zeros(1, dtype=TRI_DTYPE)
    fa, fp = _f(flat_tris)
    lib().ora_flat_tri_unpack(fp, index, out.ctypes.data)
    return out[0]


def box_test(pmin, pmax, origin, direction, tmin, tmax):
    a, ap = _f(pmin)
    b, bp = _f(pmax)
    o, op = _f(origin)
    d, dp = _f(direction)
    return bool(lib().ora_box_test(ap, bp, op, dp, tmin, tmax))


def rng_u32(seed, pixel, sample, dim):
    return lib().ora_rng_u32(seed, pixel, sample, dim)


def sincos(theta, trig):
    s, c = C.c_float(), C.c_float()
    lib().ora_sincos(theta, trig, C.byref(s), C.byref(c))
    return s.value, c.value


def camera_dir(cam, u, v):
    out = np.zeros(3, dtype=np.float32)
    lib().ora_camera_dir(C.byref(cam), u, v, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def cosine_dir(normal, u1, u2, trig):
    n, np_ = _f(normal)
    out = np.zeros(3, dtype=np.float32)
    lib().ora_cosine_dir(np_, u1, u2, trig, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


VEC_DOT, VEC_CROSS, VEC_NORMALIZE, VEC_MAT3_MUL, VEC_MIN, VEC_MAX = 0, 1, 2, 3, 4, 5


def vec_op(op, a, b=None, c=None, v=None):
    """The oracle's own vector helpers (glm's scalar definitions): dot -> float, the others -> float32[3]."""
    out = np.zeros(3, dtype=np.float32)
    keep = [_f(x) if x is not None else (None, None) for x in (a, b, c, v)]
    lib().ora_vec_op(op, keep[0][1], keep[1][1], keep[2][1], keep[3][1], out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[0] if op == VEC_DOT else out


def take_limit_counts():
    """(leaves TRAV_LIMITED skipped, leaves whose box a ray met) since the last call."""
    a, b = C.c_uint64(), C.c_uint64()
    lib().ora_take_limit_counts(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def take_leaf_replays():
    """Rays TRAV_LEAVES resolved by its ordered replay since the last call."""
    lib().ora_take_leaf_replays.restype = C.c_uint64
    return int(lib().ora_take_leaf_replays())
