"""raytracing_amd -- MI355X (gfx950) path tracer behind the plug-in interface of wcg21444/RayTracing.

The compute path is hand-written HIP in csrc/rt_hip.hip behind the C ABI of include/rt_hip.h; the C++
host side (scene models, BVH builders, plug-in adaptors) is in csrc/host.  This package is the ctypes
glue harnesses use; importing it requires the built libraries (no CPU fallback exists).
"""
from . import _native
from .scene import (MAT_DIELECTRIC, MAT_EMIT, MAT_LAMBERTIAN, MAT_METAL, ORDER_LEFT_FIRST, ORDER_RIGHT_FIRST, OldScene, SdScene,
                    device_layout)
from . import bvhdebug  # noqa: F401
from .tracer import (EST_OLD_RECURSIVE, EST_SD_LOOP, MISS, SPLIT_SAMPLE, SPLIT_TILE, CAMERA_AUTO, CAMERA_FUSED, CAMERA_SEPARATE, TRAV_AUTO, TRAV_BRUTE, TRAV_FAST_QUEUED, TRAV_FAST_INLINE, TRAV_FAST_SLICED, TRAV_FAST, TRAV_FAST_REFILL, TRAV_FAST_REPLAY, TRAV_LANE, TRAV_NESTED, TRAV_REFERENCE, TRAV_WAVE, Camera,
                     PluginHarness, Stats, Tracer, TracerError)

__all__ = ["SdScene", "OldScene", "device_layout", "Tracer", "TracerError", "Camera", "Stats", "PluginHarness", "EST_SD_LOOP",
           "EST_OLD_RECURSIVE", "ORDER_LEFT_FIRST", "ORDER_RIGHT_FIRST", "MISS", "SPLIT_TILE", "SPLIT_SAMPLE", "CAMERA_AUTO", "CAMERA_SEPARATE", "CAMERA_FUSED", "TRAV_AUTO", "TRAV_LANE", "TRAV_WAVE", "TRAV_NESTED", "TRAV_FAST", "TRAV_FAST_REPLAY", "TRAV_FAST_REFILL", "TRAV_FAST_QUEUED", "TRAV_FAST_INLINE", "TRAV_FAST_SLICED", "TRAV_BRUTE", "TRAV_REFERENCE", "MAT_LAMBERTIAN", "MAT_METAL", "MAT_EMIT", "MAT_DIELECTRIC"]
