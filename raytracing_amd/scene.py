"""Host-side scenes (built by the C++ code in csrc/host) as Python objects."""
import ctypes as C

import numpy as np

from . import _native as N

NODE_DTYPE = np.dtype([("left", "<u4"), ("right", "<u4"), ("pmin", "<f4", 3), ("pmax", "<f4", 3), ("flags", "u1"),
                       ("pad_", "u1", 3)])
TRI_DTYPE = np.dtype([("pos", "<f4", (3, 3)), ("nrm", "<f4", (3, 3)), ("uv", "<f4", (3, 2)), ("mat", "<u2"),
                      ("pad_", "<u2")])
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("mat", "<u4"), ("prim_id", "<u4")])
MATERIAL_DTYPE = np.dtype([("kind", "<u4"), ("albedo", "<f4", 4), ("param", "<f4")])
assert NODE_DTYPE.itemsize == 36 and TRI_DTYPE.itemsize == 100

MAT_LAMBERTIAN, MAT_METAL, MAT_EMIT, MAT_DIELECTRIC = 0, 1, 2, 4
ORDER_RIGHT_FIRST, ORDER_LEFT_FIRST = 0, 1


def _err():
    return N.host().rth_last_error().decode()


def _view(ptr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_uint8 * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class SdScene:
    """The reference's flat-array scene (sd::Scene): triangle meshes under a median-split BVH."""

    order = ORDER_RIGHT_FIRST  # sd::BVH::IntersectLoop pops the right child first

    def __init__(self):
        self._h = N.host().rth_sd_create()

    def __del__(self):
        if getattr(self, "_h", None):
            N.host().rth_sd_destroy(self._h)
            self._h = None

    def load_obj(self, path):
        if N.host().rth_sd_load_obj(self._h, str(path).encode()) != 0:
            raise RuntimeError(_err())
        return self

    def add_model(self, tris24, mesh_sizes=None):
        t = np.ascontiguousarray(tris24, dtype=np.float32).reshape(-1, 24)
        ms = np.ascontiguousarray(mesh_sizes if mesh_sizes is not None else [len(t)], dtype=np.uint32)
        assert int(ms.sum()) == len(t)
        if N.host().rth_sd_add_model(self._h, _fp(t), ms.ctypes.data_as(C.POINTER(C.c_uint32)), len(ms)) != 0:
            raise RuntimeError(_err())
        return self

    def finalize(self):
        if N.host().rth_sd_finalize(self._h) != 0:
            raise RuntimeError(_err())
        return self

    def counts(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        N.host().rth_sd_counts(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    @property
    def root(self):
        return self.counts()[2]

    def nodes(self):
        return _view(N.host().rth_sd_nodes(self._h), self.counts()[0], NODE_DTYPE)

    def tris(self):
        return _view(N.host().rth_sd_tris(self._h), self.counts()[1], TRI_DTYPE)

    def spheres(self):
        return np.zeros(0, dtype=SPHERE_DTYPE)

    def materials(self):
        return np.zeros(0, dtype=MATERIAL_DTYPE)

    def flat(self):
        n, t, _ = self.counts()
        a = np.zeros((n, 9), dtype=np.float32)
        b = np.zeros((t, 25), dtype=np.float32)
        N.host().rth_sd_flat(self._h, _fp(a), _fp(b))
        return a, b

    def publish(self):
        """Install as the process-global Storage::SdScene the plug-in load method reads."""
        if N.host().rth_sd_publish(self._h) != 0:
            raise RuntimeError(_err())


class OldScene:
    """The reference's pointer-based scene: spheres and meshes with per-object materials."""

    order = ORDER_LEFT_FIRST  # BVH::intersect recurses into the left child first

    def __init__(self):
        self._h = N.host().rth_old_create()

    def __del__(self):
        if getattr(self, "_h", None):
            N.host().rth_old_destroy(self._h)
            self._h = None

    def add_sphere(self, center, radius, kind, albedo=(1, 1, 1, 1), param=0.0):
        c = np.ascontiguousarray(center, dtype=np.float32)
        a = np.ascontiguousarray(albedo, dtype=np.float32)
        if N.host().rth_old_add_sphere(self._h, _fp(c), radius, kind, _fp(a), param) != 0:
            raise RuntimeError(_err())
        return self

    def add_mesh(self, tris24, kind, albedo=(1, 1, 1, 1), param=0.0):
        t = np.ascontiguousarray(tris24, dtype=np.float32).reshape(-1, 24)
        a = np.ascontiguousarray(albedo, dtype=np.float32)
        if N.host().rth_old_add_mesh(self._h, _fp(t), len(t), kind, _fp(a), param) != 0:
            raise RuntimeError(_err())
        return self

    def add_obj(self, path, kind, albedo=(1, 1, 1, 1), param=0.0):
        a = np.ascontiguousarray(albedo, dtype=np.float32)
        if N.host().rth_old_add_obj(self._h, str(path).encode(), kind, _fp(a), param) != 0:
            raise RuntimeError(_err())
        return self

    def build(self):
        if N.host().rth_old_build(self._h) != 0:
            raise RuntimeError(_err())
        return self

    finalize = build

    def counts5(self):
        v = [C.c_uint32() for _ in range(5)]
        N.host().rth_old_counts(self._h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def counts(self):
        n, t, s, m, r = self.counts5()
        return n, t, r

    @property
    def root(self):
        return self.counts5()[4]

    def nodes(self):
        return _view(N.host().rth_old_nodes(self._h), self.counts5()[0], NODE_DTYPE)

    def tris(self):
        return _view(N.host().rth_old_tris(self._h), self.counts5()[1], TRI_DTYPE)

    def spheres(self):
        return _view(N.host().rth_old_spheres(self._h), self.counts5()[2], SPHERE_DTYPE)

    def materials(self):
        return _view(N.host().rth_old_materials(self._h), self.counts5()[3], MATERIAL_DTYPE)

    def publish(self):
        if N.host().rth_old_publish(self._h) != 0:
            raise RuntimeError(_err())


def device_layout(scene):
    """What rt_upload_scene lays out for the device, as numpy arrays (host-only; rt_layout_scene): a dict with
    `fast` and `ref` node records (n, 8) float32, `tri_pos` (slots, 12) and `spheres` (slots, 8); for flat scenes also
    `grid` (n, 4) uint32, `leaf_records` (slots, 16) float32, `grid_origin` and `grid_step` (3,) float32 (`grid` is
    empty for other scenes), and `wide` (nodes, 16) uint32, the wide form of the fast tree (empty when there is none), with `has_distance_limit`,
    `leaf_class` (slots,) uint8 and `limit_constants` (3, 3) float32: rows c1, 2 c2, c2 W + c3, columns the three classes."""
    nodes, tris, sph, mats = scene.nodes(), scene.tris(), scene.spheres(), scene.materials()

    def ptr(a):
        return a.ctypes.data if len(a) else None
    v = N.LayoutViewC()
    rc = N.hip().rt_layout_scene(ptr(nodes), len(nodes), scene.root, ptr(tris), len(tris), ptr(sph), len(sph), ptr(mats),
                                 len(mats), scene.order, C.byref(v))
    if rc != 0:
        raise RuntimeError(f"rt_layout_scene failed: {rc}")

    def grab(p, rows, cols):
        if rows == 0:
            return np.zeros((0, cols), dtype=np.float32)
        return np.ctypeslib.as_array(p, shape=(rows, cols)).copy()
    out = {"fast": grab(v.fast_nodes, v.n_fast_nodes, 8), "ref": grab(v.ref_nodes, v.n_ref_nodes + 1, 8),
           "tri_pos": grab(v.tri_pos, v.n_tri_slots, 12), "spheres": grab(v.spheres, v.n_sphere_slots, 8),
           "grid": (np.ctypeslib.as_array(v.grid_nodes, shape=(v.n_grid_nodes, 4)).copy() if v.n_grid_nodes
                    else np.zeros((0, 4), dtype=np.uint32)),
           "leaf_records": grab(v.leaf_records, v.n_tri_slots if v.n_grid_nodes else 0, 16),
           "grid_origin": np.array(list(v.grid_origin), dtype=np.float32), "grid_step": np.array(list(v.grid_step), dtype=np.float32),
           "wide": (np.ctypeslib.as_array(v.wide_nodes, shape=(v.n_wide_nodes, 16)).copy() if v.n_wide_nodes
                    else np.zeros((0, 16), dtype=np.uint32)),
           "has_distance_limit": bool(v.has_distance_limit),
           "leaf_class": (np.ctypeslib.as_array(v.leaf_class, shape=(v.n_tri_slots,)).copy() if v.n_wide_nodes
                          else np.zeros(0, dtype=np.uint8)),
           "limit_constants": np.array(list(v.limit_constants), dtype=np.float32).reshape(3, 3)}
    N.hip().rt_layout_free(C.byref(v))
    return out
