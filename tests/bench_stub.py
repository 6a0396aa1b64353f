"""Stand-in for raytracing_amd.Tracer when bench.py's launcher and collectives are driven on a machine without a GPU
(RT_BENCH_TRACER=bench_stub:StubTracer with RT_BENCH_BACKEND=gloo; tests/test_bench_launcher_cpu.py).  The oracle renders
the rank's share into the host tensor bench.py hands over, as tests/test_distributed_cpu.py does: what is under test is
bench.py -- its launcher, its shares, its timing protocol and its JSON line -- not the renderer."""
import ctypes
import time

import numpy as np

from helpers import O, oracle_camera, oracle_workload_scene


class StubStats:
    def __init__(self):
        self.rays = self.box_tests = self.tri_tests = self.sph_tests = self.tri_hits = self.replays = 0
        self.kernel_ms = self.total_ms = self.kernel_ms_sum = 0.0
        self.launches = 0


class StubTracer:
    def __init__(self, device=0):
        self.stats = StubStats()
        self.opt = {}

    def bind_workload(self, wl):
        self.scene = oracle_workload_scene(wl)

    def upload(self, scene):
        pass

    def set_camera(self, cam):
        self.cam = oracle_camera(cam)

    def set_options(self, bounce_limit=4, estimator=0, seed=1, jitter=0.001, tile_rank=0, tile_world=1, collect_stats=False,
                    sample_offset=0, traversal=0, **_):
        self.opt = dict(bounce_limit=bounce_limit, seed=seed, jitter=jitter, tile_rank=tile_rank, tile_world=tile_world,
                        sample_offset=sample_offset)

    def set_pipelining(self, on):
        pass

    def reserve(self, w, h, n):
        pass

    def join(self):
        pass

    def trace_device(self, w, h, first, n, ptr):
        image = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), shape=(h, w, 4))
        t0 = time.perf_counter()
        _, _, st = self.scene.render(self.cam, O.make_options(threads=2, **self.opt), w, h, first, n, image=image)
        s = self.stats
        s.rays += st.rays
        s.box_tests += st.box_tests
        s.tri_tests += st.tri_tests
        s.tri_hits += st.tri_hits
        s.kernel_ms_sum += (time.perf_counter() - t0) * 1e3
        s.launches += 1

    def sync(self):
        out, self.stats = self.stats, StubStats()
        return out

    def close(self):
        pass
