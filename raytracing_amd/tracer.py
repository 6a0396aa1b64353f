"""Python front end of the C ABI in include/rt_hip.h (ctypes; no torch types cross the boundary)."""
import ctypes as C

import numpy as np

from . import _native as N

EST_SD_LOOP, EST_OLD_RECURSIVE = 0, 1
TRAV_AUTO, TRAV_LANE, TRAV_WAVE, TRAV_NESTED, TRAV_FAST, TRAV_FAST_REPLAY, TRAV_REFERENCE, TRAV_FAST_REFILL, TRAV_BRUTE, TRAV_FAST_QUEUED, TRAV_FAST_INLINE, TRAV_FAST_SLICED = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
MISS = 0xFFFFFFFF
SPLIT_TILE, SPLIT_SAMPLE = 0, 1
CAMERA_AUTO, CAMERA_SEPARATE, CAMERA_FUSED = 0, 1, 2

_STATUS = {0: "RT_OK", -1: "RT_ERR_INVALID_ARGUMENT", -2: "RT_ERR_NO_DEVICE", -3: "RT_ERR_HIP", -4: "RT_ERR_NO_SCENE",
           -5: "RT_ERR_BAD_SCENE", -6: "RT_ERR_OUT_OF_MEMORY"}


class TracerError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"{_STATUS.get(code, code)}: {text}")
        self.code = code


class Camera:
    """Fields of the reference's Camera (Include/Camera.hpp:10-31); height = width / aspect."""

    def __init__(self, focal_length=1.0, position=(0.0, 0.0, 1.0), width=2.0, aspect_ratio=16.0 / 9.0,
                 look_at=(0.0, 0.0, 0.0)):
        self.focal_length = float(focal_length)
        self.position = tuple(float(x) for x in position)
        self.look_at = tuple(float(x) for x in look_at)
        self.width = float(width)
        self.aspect_ratio = float(aspect_ratio)
        self.height = float(np.float32(width) / np.float32(aspect_ratio))

    def to_c(self):
        c = N.CameraC()
        c.focal_length = self.focal_length
        c.position[:] = self.position
        c.look_at[:] = self.look_at
        c.width, c.height, c.aspect_ratio = self.width, self.height, self.aspect_ratio
        return c

    def as_floats(self):
        return np.array([self.focal_length, *self.position, *self.look_at, self.width, self.height,
                         self.aspect_ratio], dtype=np.float32)


class Stats:
    def __init__(self, c):
        self.rays, self.box_tests, self.tri_tests = c.rays, c.box_tests, c.tri_tests
        self.sph_tests, self.tri_hits, self.replays = c.sph_tests, c.tri_hits, c.replays
        self.kernel_ms, self.total_ms = c.kernel_ms, c.total_ms
        self.kernel_ms_sum, self.launches = c.kernel_ms_sum, c.launches

    def __repr__(self):
        return (f"Stats(rays={self.rays}, box={self.box_tests}, tri={self.tri_tests}, sph={self.sph_tests}, "
                f"tri_hits={self.tri_hits}, kernel_ms={self.kernel_ms:.3f}, total_ms={self.total_ms:.3f})")


class Tracer:
    """One device context (rt_ctx)."""

    def __init__(self, device=0, devices=None):
        """device: one HIP device; devices: a list of HIP devices behind one context (rt_create_multi)."""
        self._ctx = C.c_void_p()
        if devices is not None:
            ids = (C.c_int * len(devices))(*devices)
            rc = N.hip().rt_create_multi(C.byref(self._ctx), ids, len(devices))
        else:
            rc = N.hip().rt_create(C.byref(self._ctx), device)
        if rc != 0:
            self._ctx = None
            raise TracerError(rc, "rt_create: no usable HIP device (the tracer has no CPU fallback)")
        self._keep = None

    def close(self):
        if getattr(self, "_ctx", None):
            N.hip().rt_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise TracerError(rc, N.hip().rt_last_error(self._ctx).decode())

    def upload(self, scene):
        nodes, tris, sph, mats = scene.nodes(), scene.tris(), scene.spheres(), scene.materials()
        self.upload_arrays(nodes, scene.root, tris, sph, mats, scene.order)

    def upload_arrays(self, nodes, root, tris, spheres, mats, order):
        def ptr(a):
            return a.ctypes.data if len(a) else None
        self._check(N.hip().rt_upload_scene(self._ctx, ptr(nodes), len(nodes), root, ptr(tris), len(tris),
                                            ptr(spheres), len(spheres), ptr(mats), len(mats), order))

    def set_camera(self, cam):
        c = cam.to_c()
        self._check(N.hip().rt_set_camera(self._ctx, C.byref(c)))

    def set_options(self, bounce_limit=4, estimator=EST_SD_LOOP, seed=1, jitter=0.001, tile_rank=0, tile_world=1,
                    collect_stats=False, sample_offset=0, traversal=0, multi_split=0, camera_stage=0):
        o = N.OptionsC(bounce_limit, estimator, seed, jitter, tile_rank, tile_world, int(collect_stats),
                       sample_offset, traversal, multi_split, camera_stage)
        self._check(N.hip().rt_set_options(self._ctx, C.byref(o)))

    def reset(self):
        self._check(N.hip().rt_reset(self._ctx))

    def trace(self, width, height, first_sample=1, n_samples=1, want_image=True, want_primary=False):
        img = np.zeros((height, width, 4), dtype=np.float32) if want_image else None
        prim = np.zeros((height, width), dtype=np.uint32) if want_primary else None
        st = N.StatsC()
        self._check(N.hip().rt_trace(self._ctx, width, height, first_sample, n_samples,
                                     img.ctypes.data_as(C.POINTER(C.c_float)) if want_image else None,
                                     prim.ctypes.data_as(C.POINTER(C.c_uint32)) if want_primary else None,
                                     C.byref(st)))
        return img, prim, Stats(st)

    def trace_host(self, width, height, first_sample, n_samples, host_rgba_ptr):
        """rt_trace into caller-owned host memory (e.g. page-locked, alloc_host); returns the call's Stats."""
        st = N.StatsC()
        self._check(N.hip().rt_trace(self._ctx, width, height, first_sample, n_samples,
                                     C.cast(host_rgba_ptr, C.POINTER(C.c_float)), None, C.byref(st)))
        return Stats(st)

    @staticmethod
    def alloc_host(n_bytes):
        p = N.hip().rt_alloc_host(n_bytes)
        if not p:
            raise MemoryError("rt_alloc_host failed")
        return p

    @staticmethod
    def free_host(ptr):
        N.hip().rt_free_host(ptr)

    def trace_begin(self, width, height, first_sample, n_samples, host_rgba_ptr):
        self._check(N.hip().rt_trace_begin(self._ctx, width, height, first_sample, n_samples, host_rgba_ptr))

    def trace_end(self):
        st = N.StatsC()
        self._check(N.hip().rt_trace_end(self._ctx, C.byref(st)))
        return Stats(st)

    def trace_discard(self):
        self._check(N.hip().rt_trace_discard(self._ctx))

    def trace_device(self, width, height, first_sample, n_samples, d_rgba_ptr, d_primary_ptr=None):
        self._check(N.hip().rt_trace_device(self._ctx, width, height, first_sample, n_samples, d_rgba_ptr,
                                            d_primary_ptr))

    def trace_device_noalloc(self, width, height, first_sample, n_samples, d_rgba_ptr, d_primary_ptr=None):
        self._check(N.hip().rt_trace_device_noalloc(self._ctx, width, height, first_sample, n_samples, d_rgba_ptr,
                                                    d_primary_ptr))

    def reserve(self, width, height, n_samples):
        self._check(N.hip().rt_reserve(self._ctx, width, height, n_samples))

    def device_count(self):
        return N.hip().rt_device_count(self._ctx)

    def set_stream(self, stream_ptr):
        self._check(N.hip().rt_set_stream(self._ctx, stream_ptr))

    def set_pipelining(self, on):
        self._check(N.hip().rt_set_pipelining(self._ctx, 1 if on else 0))

    def join(self):
        self._check(N.hip().rt_join(self._ctx))

    def sync(self):
        st = N.StatsC()
        self._check(N.hip().rt_sync(self._ctx, C.byref(st)))
        return Stats(st)


class PluginHarness:
    """Drives the C++ plug-in adaptors (RenderPipeline<...HIPContext, Load...HIP, Trace...HIP>) the way the
    reference's Renderer::render drives a render mode: load once, then one trace call per frame."""

    MODE_SD, MODE_OLD = 0, 1

    def __init__(self, mode, width, height, cam, bounce_limit=4, seed=1, jitter=0.001, device=0, devices=None):
        f = cam.as_floats()
        if devices is not None:
            ids = (C.c_int * len(devices))(*devices)
            self._h = N.host().rth_harness_create_devices(mode, ids, len(devices), width, height,
                                                          f.ctypes.data_as(C.POINTER(C.c_float)), bounce_limit, seed, jitter)
        else:
            self._h = N.host().rth_harness_create(mode, device, width, height, f.ctypes.data_as(C.POINTER(C.c_float)),
                                                  bounce_limit, seed, jitter)
        if not self._h:
            raise RuntimeError(N.host().rth_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            N.host().rth_harness_destroy(self._h)
            self._h = None

    __del__ = close

    def _call(self, fn, *a):
        if fn(self._h, *a) != 0:
            raise RuntimeError(N.host().rth_last_error().decode())

    def set_look_ahead(self, on):
        N.host().rth_harness_set_look_ahead(self._h, 1 if on else 0)

    def load(self):
        self._call(N.host().rth_harness_load)

    def render(self):
        self._call(N.host().rth_harness_render)

    def reset(self):
        self._call(N.host().rth_harness_reset)

    def resize(self, w, h):
        self._call(N.host().rth_harness_resize, w, h)

    def set_camera_position(self, pos):
        p = np.ascontiguousarray(pos, dtype=np.float32)
        N.host().rth_harness_set_camera_position(self._h, p.ctypes.data_as(C.POINTER(C.c_float)))

    @staticmethod
    def set_bvh_accel(on):
        """BVHSettings::toggleBVHAccel (Program/BVHUI.hpp:14), a process-wide switch as in the reference; the UI sets
        RenderState::Dirty with it (BVHUI.hpp:25), i.e. the caller resets the samples."""
        N.host().rth_set_bvh_accel(1 if on else 0)

    def sample_count(self):
        return N.host().rth_harness_sample_count(self._h)

    def last_rays(self):
        """Rays the trace method counted for the last frame (its lastStats)."""
        return N.host().rth_harness_last_rays(self._h)

    def output(self):
        w, h = C.c_uint32(), C.c_uint32()
        N.host().rth_harness_output(self._h, None, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value, 4), dtype=np.float32)
        N.host().rth_harness_output(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(w), C.byref(h))
        return out
