#!/usr/bin/env python3
"""Throughput benchmark of the HIP path tracer: Mrays/s (all bounces), one JSON line on rank 0.

A step is one trace pass of `spp` samples per pixel over the whole frame of the workload (the batch: W x H x spp pixel
samples).  The K timed steps accumulate progressively into one running-mean image on the device, as the reference's
frame loop does (Renderer/TracerImpl.cpp:9-12), and the timed region ends with the copy of the W*H*16-byte mean to
page-locked host memory on rank 0 (the reference's setData upload, GLShader/Texture.cpp:86-91).  With N > 1 the same K steps
are shared by N ranks, one process per GPU -- interleaved 8x8 image tiles (the default) or sample-index ranges -- and the
render ends with ONE RCCL reduce of the W*H*4 float image onto rank 0 in front of that copy, inside the timed region:
total work is fixed, "scaling": "strong".  `python bench.py --gpus N` starts the N ranks itself (fresh child processes,
before this process has touched a GPU) -- the counterpart of the reference's in-call fan-out over 16 row bands
(Renderer/TraceMethods.cpp:61-78); under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E nominal
VALU_PEAK_TOPS = 78.6         # 256 CU x 128 lanes x 2.4 GHz, one op per lane-cycle (the path forbids FMA contraction)
L2_BYTES = 32 << 20           # 8 XCDs x 4 MB: scenes whose node + triangle-position records fit are served from the L2s
MALL_BYTES = 256 << 20        # Infinity Cache: scenes up to this size are served on-die as well (measured: the 133 MB field)


def host_cpu():
    """Model string, logical CPUs of the machine, CPUs this process may use, and the thread count the harness asks
    worker pools to keep to (OMP_NUM_THREADS on the GPU box: the box's CPU share for one GPU)."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    return model, logical, usable, share


def oracle_scene(wl):
    """The workload's scene rebuilt for the CPU oracle (cpu_baseline leg only)."""
    from oracle import oracle as O
    import raytracing_amd as R
    if wl.estimator == R.EST_SD_LOOP:
        osc = O.SdScene()
        for tris in wl.models:
            osc.add_model(tris, [len(tris)])
        osc.finalize()
        return osc
    osc = O.OldScene()
    for c, rad, kind, alb, param in wl.spheres:
        osc.add_sphere(c, rad, kind, alb, param)
    osc.build()
    return osc


def cpu_baseline(wl, args):
    """Times the CPU oracle (a port of the reference path) on a bounded sample of the workload: the same scene, camera
    and bounce limit at reduced resolution and a few spp (about 10-20 s of CPU work), (ii) one row band per host thread
    the process may use and (i) the reference's own schedule of 16 bands with a lock per pixel."""
    from oracle import oracle as O
    from raytracing_amd import workloads as WL
    import raytracing_amd as R
    model, logical, usable, share = host_cpu()
    threads = min(usable, share) if share > 0 else usable
    if args.cpu_threads:
        threads = args.cpu_threads
    cam = wl.camera
    ocam = O.make_camera(cam.focal_length, cam.position, cam.look_at, cam.width, cam.aspect_ratio)
    osc = oracle_scene(wl)
    # (polynomial cos/sin so that its visit counts can be compared with the GPU's; the cost of the two
    # trig evaluations per bounce is negligible next to the traversal)
    opt = O.make_options(bounce_limit=wl.bounce_limit, seed=wl.seed, jitter=wl.jitter, trig=O.TRIG_POLY,
                         threads=threads, lock_per_pixel=0, accum=O.ACCUM_FORWARD)
    # The sample is sized from a probe (1 spp at 1/32 of the resolution) so that each of the two timed renders is about
    # args.cpu_seconds of CPU work whatever the workload costs per ray: the largest resolution among 1/4 ... 1/32 at
    # which one sample per pixel fits, and as many samples per pixel as fit.
    pw, ph = max(wl.width // 32, 16), max(wl.height // 32, 9)
    _, _, probe = osc.render(ocam, opt, pw, ph, 1, 1)
    per_pixel = max(probe.seconds, 1e-5) / (pw * ph)
    div = args.cpu_div
    while div < 32 and per_pixel * (wl.width // div) * (wl.height // div) > args.cpu_seconds:
        div *= 2
    w, h = max(wl.width // div, 16), max(wl.height // div, 9)
    if per_pixel * w * h < 0.25 * args.cpu_seconds:  # cheap rays: the small probe mostly measured thread start-up; probe again at size
        _, _, probe = osc.render(ocam, opt, w, h, 1, 1)
        per_pixel = max(probe.seconds, 1e-5) / (w * h)
    spp = args.cpu_spp or int(min(max(args.cpu_seconds / (per_pixel * w * h), 1), 4096))
    args.cpu_spp = spp
    _, _, st = osc.render(ocam, opt, w, h, 1, spp)
    # (i) the reference's own schedule: 16 row bands per sample and a shared_lock per pixel (TraceMethods.cpp:57, :61-78)
    opt_ref = O.make_options(bounce_limit=wl.bounce_limit, seed=wl.seed, jitter=wl.jitter, trig=O.TRIG_LIBM,
                             threads=16, lock_per_pixel=1, accum=O.ACCUM_RECURSIVE)
    _, _, st_ref = osc.render(ocam, opt_ref, w, h, 1, spp)
    return {"value": st.rays / st.seconds / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{wl.name} scene/camera/bounce limit at {w}x{h}, {args.cpu_spp} spp ({st.rays} rays, {st.seconds:.2f} s)",
            "host_cpu_model": model, "host_logical_cpus": logical, "cpus_usable_by_this_process": usable,
            "thread_cap_of_the_box": share or None,
            "reference_schedule_value": st_ref.rays / st_ref.seconds / 1e6,
            "reference_schedule": "16 std::async row bands + shared_lock per pixel (TraceMethods.cpp:57, :61-78)"}, st, (w, h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dragon18520_1080p")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per step (0 = the workload's bench_spp)")
    ap.add_argument("--split", choices=("tile", "sample"), default="tile",
                    help="N > 1: tile = the ranks share the 8x8 tiles of every step (the default); "
                         "sample = the ranks share the samples of every step")
    ap.add_argument("--traversal", type=int, default=4, help="RT_TRAV_* of the timed steps (4 = fast walk, 7 = with lane re-fill)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-counts", action="store_true",
                    help="skip the untimed pass that counts the tests of the reference-order walk")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed primary-only and host-copy legs")
    ap.add_argument("--cpu-div", type=int, default=4, help="cpu_baseline sample: resolution divisor")
    ap.add_argument("--cpu-spp", type=int, default=0, help="cpu_baseline sample: samples per pixel (0 = sized from a probe)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="cpu_baseline sample: target seconds per timed render")
    ap.add_argument("--cpu-threads", type=int, default=0, help="cpu_baseline: threads of schedule (ii) (0 = what the process may use)")
    args = ap.parse_args()
    if os.environ.get("RT_BENCH_WATCHDOG"):  # dump every thread's stack if the run is still going after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RT_BENCH_WATCHDOG"]), exit=True)

    import torch
    import raytracing_amd as R
    from raytracing_amd import workloads as WL

    import torch.distributed as dist
    from raytracing_amd import parallel as P

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the tracer has no CPU path")
    # RT_BENCH_BACKEND=gloo lets the N > 1 path be rehearsed with several ranks sharing one GPU (RCCL refuses that)
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend != "nccl":
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(n_dev, 1))
    rank, world, local_rank = P.init_from_env(backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def note(msg):  # progress on stderr (a multi-minute run must not look hung)
        print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    wl = WL.make_workload(args.workload)
    spp = args.spp or wl.bench_spp
    W, H = wl.width, wl.height
    split = args.split
    if split == "sample" and spp % world:
        raise SystemExit("--split sample needs spp divisible by the number of GPUs")
    my_spp = spp // world if split == "sample" else spp          # samples per pixel this rank renders per step
    my_pixels = W * H                                            # ... of how many pixels (tile split: its tiles')
    tr = R.Tracer(local_rank)
    tr.upload(wl.scene)
    tr.set_camera(wl.camera)
    tr.set_stream(torch.cuda.current_stream().cuda_stream)
    img = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    share = P.share_options(split, rank, world, samples_per_rank=my_spp * max(args.steps, args.warmup, 1))

    def options(stats, traversal=args.traversal, bounce_limit=wl.bounce_limit):
        tr.set_options(bounce_limit=bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter,
                       collect_stats=stats, traversal=traversal, **share)

    # The steps of a render are issued back to back with pipelining on: the tracer overlaps the latency-bound end of one
    # chunk of samples (deep bounce levels) with the start of the next on its own streams; rt_join orders the image
    # before the reduce.
    options(False)
    tr.set_pipelining(True)
    tr.reserve(W, H, my_spp)      # set-up, untimed like the upload: the timed region never allocates

    def trace_pass(i):
        tr.trace_device(W, H, 1 + i * my_spp, my_spp, img.data_ptr())

    def render(n):
        P.render(trace_pass, tr.join, img, n, split, world)

    # work of one step: counted once, untimed, by the counting variant of the same kernels
    options(True)
    trace_pass(0)
    counts = tr.sync()
    # the same rays walked in the reference's own visiting order over the reference's own boxes: what the reference
    # traversal (and the CPU baseline) has to do per ray -- on 16 spp (that walk is slow), skipped for scenes where it
    # would take minutes (the million-triangle fields)
    ref_counts = fast16 = None
    ref_spp = min(my_spp, 16)
    if rank == 0 and not args.no_reference_counts and len(wl.scene.nodes()) < 200000:
        options(True)
        tr.trace_device(W, H, 1, ref_spp, img.data_ptr())
        fast16 = tr.sync()
        options(True, R.TRAV_REFERENCE)
        tr.trace_device(W, H, 1, ref_spp, img.data_ptr())
        ref_counts = tr.sync()
        assert ref_counts.rays == fast16.rays and ref_counts.tri_hits == fast16.tri_hits
    options(False)
    note(f"scene uploaded, work counted ({counts.rays} rays per step)")

    render(args.warmup)
    tr.sync()
    note("warm-up done")
    grouped = dist.is_initialized()   # N > 1, or a single rank taken through the collectives (RT_FORCE_DIST=1)
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    render(args.steps)
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    note(f"{args.steps} timed steps in {elapsed * 1e3:.2f} ms")
    st = tr.sync()
    assert st.launches == args.steps  # st.rays: closest-hit searches of all the timed steps of this rank

    total = torch.tensor([elapsed, float(st.rays)], dtype=torch.float64, device=dev)
    if grouped:
        tmax = total.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        elapsed, rays_total = float(tmax[0]), float(total[1])
    else:
        rays_total = float(total[1])

    if rank == 0:
        kernel_ms = st.kernel_ms_sum / st.launches   # HIP events on the tracer's streams: first start to last end / steps
        if split == "tile" and world > 1:
            my_pixels = sum(1 for k in range(((W + 7) // 8) * ((H + 7) // 8)) if k % world == rank) * 64

        flat = wl.estimator == R.EST_SD_LOOP  # walked over the 16-byte grid records, leaves tested from 64-byte records

        def traversal_bytes(c):
            # flat scenes: 16 B per slab test (grid box + link), 64 B per leaf test (the leaf's exact box and its
            # triangle in one record); pointer scenes: 32 B per slab test, 36 B per triangle test (3 positions), 16 B
            # per sphere test; both: 36 B of normals per traversal that ends on a triangle.  The counts are those of the
            # exact walk (the statistics variant of the kernels); the grid walk makes a few per cent more slab tests.
            if flat:
                return 16 * c.box_tests + 64 * c.tri_tests + 36 * c.tri_hits
            return 32 * c.box_tests + 36 * c.tri_tests + 16 * c.sph_tests + 36 * c.tri_hits

        def traversal_ops(c):
            return 27 * c.box_tests + 56 * c.tri_tests + 24 * c.sph_tests + 60 * c.rays

        # per pixel: the 16 B mean read and written once per step
        image_bytes = 32 * my_pixels
        alg_bytes = traversal_bytes(counts) + image_bytes
        alg_ops = traversal_ops(counts)
        secs = kernel_ms * 1e-3
        # hardware-counter figures of this workload, measured in separate rocprofv3 --pmc passes and kept with the
        # profiles: HBM bytes per step (traffic), VALU issue utilisation, active lanes per VALU instruction
        pmc = {}
        ppath = os.path.join(ROOT, "profiles", "pmc_per_step.json")
        if os.path.exists(ppath):
            with open(ppath) as fh:
                pmc = json.load(fh).get(f"{wl.name}:{my_spp}", {})
        traffic = pmc.get("hbm_bytes_per_step")
        scene_bytes = wl.scene_bytes()
        on_die = scene_bytes <= MALL_BYTES
        if on_die:
            # node and triangle records are served by the L2s (<= 32 MB) or the Infinity Cache (<= 256 MB): HBM carries the
            # queues and the image only; the roof the walk is priced against is VALU arithmetic (what binds it in practice is the
            # vector-memory path: valu_issue_utilisation and the texture-addresser figure of the counter passes say how far)
            roof = {"bound": "valu", "achieved": alg_ops / secs / 1e12, "peak": VALU_PEAK_TOPS, "unit": "Tops/s",
                    "frac": alg_ops / secs / 1e12 / VALU_PEAK_TOPS, "traffic": traffic,
                    "algorithmic_ops_per_launch": int(alg_ops),
                    "note": "fp32 lane-operations the walk's own tests and the shading need (27 per slab test, 56 per "
                            "triangle test, 24 per sphere test, 60 per ray) against the non-FMA VALU peak; the scene's records ("
                            + "%.1f MB) are served by the %s, so HBM carries only the queues and the image (hbm_*)"
                            % (scene_bytes / 1e6, "L2s" if scene_bytes <= L2_BYTES else "Infinity Cache")}
        else:
            roof = {"bound": "hbm", "achieved": alg_bytes / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": alg_bytes / secs / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(alg_bytes),
                    "note": ("bytes the walk's own tests need (%s, 36 of normals per triangle hit) + the image; the scene's "
                             "records (%.0f MB) exceed every cache"
                             % ("16 per slab test on the grid records, 64 per leaf test" if flat
                                else "32 per slab test, 36 per triangle test, 16 per sphere test", scene_bytes / 1e6))}
        roof["valu_frac"] = alg_ops / secs / 1e12 / VALU_PEAK_TOPS
        roof.update({"kernel": "one step = one trace call: flat scenes k_camera + k_sliced for the rounds that fill the chip + "
                               "k_refill for the rest (short-walk scenes: k_refill throughout); pointer scenes k_path, or "
                               "k_refill beyond the L2s; one launch per round / bounce level and chunk of samples, + k_fold",
                     "kernel_ms": kernel_ms,
                     "hbm_achieved_GBps": traffic / secs / 1e9 if traffic else None,
                     "hbm_frac": traffic / secs / 1e9 / HBM_PEAK_GBS if traffic else None,
                     "valu_issue_utilisation": pmc.get("valu_issue_utilisation"),
                     "active_lanes_per_valu_instruction": pmc.get("active_lanes"),
                     # what the walk kernels run against in practice: every lane of a wave fetches a record of its own, and a
                     # compute unit serves such loads at 1.98 cache lines per clock out of its L1 and 0.437 on an L1 miss
                     # (tools/ta_probe.hip, profiles/r02_ta_probe.txt).  floor = the step's counted line accesses and misses
                     # (TCP_TOTAL_CACHE_ACCESSES, TCP_TCC_READ_REQ) at those rates; frac = floor / cycles of the step
                     "vector_memory_path": pmc.get("vector_memory_path"),
                     "counters_from": pmc.get("source")})
        primary = my_pixels * my_spp if wl.bounce_limit > 0 or wl.estimator != R.EST_SD_LOOP else 0
        out = {
            "metric": "Mrays/s (all bounces)", "value": rays_total / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "msamples_per_s": (W * H * spp * args.steps) / elapsed / 1e6,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "Stanford Dragon mesh bundled with the reference (data/models), no external data"
                    if wl.estimator == R.EST_SD_LOOP else "synthetic",
            "config": {"workload": wl.name, "description": wl.description, "width": W, "height": H, "spp_per_step": spp,
                       "bounce_limit": wl.bounce_limit, "triangles": int(len(wl.scene.tris())),
                       "rays_per_step_per_gpu": int(st.rays // args.steps),
                       "traversal": "fast walk (own hierarchy over the reference's leaf boxes; hits bit-identical "
                                    "to the reference-order walk), primitive tests noted and run 64 at a time",
                       "parallelism": "single GPU" if world == 1 else
                                      "scene replicated, interleaved 8x8 image tiles per GPU (tile k -> GPU k % N), steps accumulate on "
                                      "the device, one RCCL reduce of the zero-padded W*H*4 float image at the end of the render"
                                      if split == "tile" else
                                      "scene replicated, sample-index range per GPU, steps accumulate on the device, one RCCL "
                                      "reduce of the W*H*4 float mean at the end of the render"},
            "roofline": roof,
            "rays_per_step": {"primary": int(primary), "bounce": int(counts.rays - primary),
                              "searches_ending_on_a_triangle": int(counts.tri_hits),
                              "note": "primary = camera rays; a camera ray that misses the scene ends its path at once"},
            "counts_per_launch": {"rays": int(counts.rays), "box_tests": int(counts.box_tests),
                                  "tri_tests": int(counts.tri_tests), "sph_tests": int(counts.sph_tests),
                                  "tri_hits": int(counts.tri_hits), "replays": int(counts.replays)},
        }
        if ref_counts is not None:
            out["reference_order_equivalent"] = {
                "spp": ref_spp, "rays": int(ref_counts.rays),
                "box_tests_per_ray": ref_counts.box_tests / max(ref_counts.rays, 1),
                "tri_tests_per_ray": ref_counts.tri_tests / max(ref_counts.rays, 1),
                "fast_walk_box_tests_per_ray": fast16.box_tests / max(fast16.rays, 1),
                "note": "tests the reference's own traversal performs for the same rays and the same hits"}
        if world == 1 and not args.no_extras:
            note("extra legs: camera rays alone, host-image boundary")
            # (1) camera rays alone: the same steps with bounce limit 1 (every path ends after its first search)
            options(False, bounce_limit=1)
            ksteps = max(min(args.steps, 8), 2)
            render(1)
            tr.sync()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            render(ksteps)
            torch.cuda.synchronize()
            t_primary = (time.perf_counter() - t1) / ksteps
            pst = tr.sync()
            t_step = elapsed / args.steps
            out["rays_per_step"].update({
                "primary_only_ms_per_step": t_primary * 1e3,
                "primary_Mrays_per_s": pst.rays / ksteps / t_primary / 1e6,
                "bounce_Mrays_per_s": (counts.rays - primary) / max(t_step - t_primary, 1e-9) / 1e6 if counts.rays > primary else None,
                "how": "primary: the same steps with bounce limit 1; bounce: the remaining rays over the remaining time"})
            # (2) through the synchronous host-image boundary: rt_trace into page-locked memory, D2H of the mean included
            options(False)
            tr.set_pipelining(False)
            hbuf = tr.alloc_host(W * H * 16)
            tr.trace_host(W, H, 1, my_spp, hbuf)
            t2 = time.perf_counter()
            hrays = 0
            for i in range(ksteps):
                hrays += tr.trace_host(W, H, 1 + i * my_spp, my_spp, hbuf).rays
            t_host = (time.perf_counter() - t2) / ksteps
            tr.free_host(hbuf)
            out["incl_d2h"] = {"value": hrays / ksteps / t_host / 1e6, "unit": "Mrays/s", "ms_per_step": t_host * 1e3,
                               "note": "rt_trace, one synchronous call per step, the W*H*16-byte mean copied to page-locked "
                                       "host memory inside the timed call (the reference's setData upload, Texture.cpp:86-91)"}
        if world == 1 and not args.no_cpu_baseline:
            note("CPU baseline (two renders of about %.0f s each)" % args.cpu_seconds)
            cb, ost, (cw, ch) = cpu_baseline(wl, args)
            note("CPU baseline done; checking its test counts against the GPU's reference-order walk")
            # same bounded sample on the GPU: counts must equal the oracle's (checker, not the thing measured)
            tr.set_pipelining(False)
            tr.set_options(bounce_limit=wl.bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter,
                           collect_stats=True, traversal=R.TRAV_REFERENCE)
            _, _, gst = tr.trace(cw, ch, 1, args.cpu_spp, want_image=False)
            cb["gpu_counts_match_oracle"] = bool((gst.rays, gst.box_tests, gst.tri_tests, gst.sph_tests) ==
                                                 (ost.rays, ost.box_tests, ost.tri_tests, ost.sph_tests))
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    tr.close()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
