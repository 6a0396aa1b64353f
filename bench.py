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


def source_stamp():
    """What the counter figures under profiles/ were measured on is recorded there as a hash of the kernel sources and of
    the built library; this is the same pair for the tree this run loads."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "raytracing_amd", "csrc")
    names = sorted(n for n in os.listdir(csrc) if n.endswith((".hip", ".hpp")))
    for path in [os.path.join(csrc, n) for n in names] + [os.path.join(ROOT, "include", "rt_hip.h")]:
        with open(path, "rb") as fh:
            h.update(fh.read())
    lib = os.path.join(ROOT, "raytracing_amd", "librt_hip.so")
    so = None
    if os.path.exists(lib):
        with open(lib, "rb") as fh:
            so = hashlib.sha256(fh.read()).hexdigest()[:16]
    return {"kernel_sources_sha256": h.hexdigest()[:16], "librt_hip_sha256": so}


def spawn_ranks(args, argv):
    """`bench.py --gpus N` on its own: N fresh child processes, one per GPU, started BEFORE this process makes any GPU call
    (it only counts devices); each runs this file as rank r of N, rank 0 prints the JSON line, this process waits for all
    of them and hands their exit status on.  The counterpart of the reference's fan-out inside the trace call
    (16 std::async row bands, Renderer/TraceMethods.cpp:61-78): the caller launches nothing."""
    n = args.gpus
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        import torch
        visible = torch.cuda.device_count()   # counts devices; does not initialise the GPU
        if visible < n:
            print(f"bench.py: --gpus {n} asked for, {visible} HIP device(s) visible: not measuring fewer GPUs under that label",
                  file=sys.stderr)
            return 2
    with socket.socket() as sk:   # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    limit = time.time() + float(os.environ.get("RT_BENCH_SPAWN_TIMEOUT", "1500"))
    rc = 0
    pending = list(procs)
    while pending and rc == 0:
        for pr in list(pending):
            code = pr.poll()
            if code is not None:
                pending.remove(pr)
                if code != 0:
                    rc = code if code > 0 else 1
        if time.time() > limit:
            print("bench.py: the ranks did not finish in time", file=sys.stderr)
            rc = 3
        if pending and rc == 0:
            time.sleep(0.05)
    for pr in pending:   # a rank failed (or time ran out): end the others, by their own process ids
        pr.terminate()
    for pr in pending:
        try:
            pr.wait(timeout=20)
        except subprocess.TimeoutExpired:
            pr.kill()
    return rc


def load_tracer_class():
    """The renderer of a rank.  RT_BENCH_TRACER=module:Class swaps in another class with the Tracer's interface -- the CPU
    tests drive the launcher and the collectives with a stand-in from tests/ and the gloo backend, no GPU in the box."""
    spec = os.environ.get("RT_BENCH_TRACER")
    if not spec:
        import raytracing_amd as R
        return R.Tracer, False
    import importlib
    mod, cls = spec.split(":")
    return getattr(importlib.import_module(mod), cls), True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dragon18520_1080p")
    ap.add_argument("--spp", type=int, default=0,
                    help="samples per pixel per step (0 = the workload's bench_spp, raised until the timed steps last --min-seconds)")
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="with --spp 0: samples per step are raised until the timed region lasts about this long")
    ap.add_argument("--split", choices=("auto", "tile", "sample"), default="auto",
                    help="N > 1: tile = the ranks share the 8x8 tiles of every step; sample = the ranks share the samples of "
                         "every step; auto = sample for the million-triangle fields (configuration 5), tile otherwise")
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
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    # The one JSON line is all that goes to this process's standard output: libraries print banners there while they
    # initialise (RCCL its version, gloo its connections), so file descriptor 1 points at standard error from here on and the
    # line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("RT_BENCH_WATCHDOG"):  # dump every thread's stack if the run is still going after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RT_BENCH_WATCHDOG"]), exit=True)

    import torch
    import raytracing_amd as R
    from raytracing_amd import workloads as WL

    import torch.distributed as dist
    from raytracing_amd import parallel as P

    Tracer, stand_in = load_tracer_class()
    on_gpu = not stand_in
    if on_gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the tracer has no CPU path")
    # RT_BENCH_BACKEND=gloo lets the N > 1 path be rehearsed with several ranks sharing one GPU (RCCL refuses that)
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count() if on_gpu else 0
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus and not (env_world == 1 and args.gpus == 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch one rank per GPU "
                         "(torch.distributed.run --nproc-per-node == --gpus), or leave the launching to bench.py")
    if on_gpu and backend == "nccl" and n_dev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} asked for, {n_dev} HIP device(s) visible")
    if backend != "nccl" and on_gpu:
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(n_dev, 1))
    rank, world, local_rank = P.init_from_env(backend)
    if on_gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")

    def device_sync():
        if on_gpu:
            torch.cuda.synchronize()

    def note(msg):  # progress on stderr (a multi-minute run must not look hung)
        print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    wl = WL.make_workload(args.workload)
    W, H = wl.width, wl.height
    split = args.split
    if split == "auto":   # configuration 5 (the million-triangle fields) is stated with the sample-range split
        split = "sample" if args.workload.startswith("dragon_field") else "tile"
    spp = args.spp or wl.bench_spp
    tr = Tracer(local_rank)
    if stand_in:
        tr.bind_workload(wl)
    tr.upload(wl.scene)
    tr.set_camera(wl.camera)
    if on_gpu:
        tr.set_stream(torch.cuda.current_stream().cuda_stream)
    img = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    # where the render's mean lands on the host (rank 0): page-locked, so the copy runs at the link's rate
    host_img = torch.empty((H, W, 4), dtype=torch.float32, pin_memory=on_gpu) if rank == 0 else None
    state = {"spp": spp, "my_spp": 0, "share": {}, "weight": None}

    def options(stats, traversal=args.traversal, bounce_limit=wl.bounce_limit):
        tr.set_options(bounce_limit=bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter,
                       collect_stats=stats, traversal=traversal, **state["share"])

    def set_spp(new_spp):
        """Samples per pixel of a step; this rank's part of them and of the RNG's sample indices."""
        state["spp"] = new_spp
        state["my_spp"] = P.samples_of_rank(new_spp, rank, world) if split == "sample" else new_spp
        state["share"] = P.share_options(split, rank, world, spp=new_spp, steps=max(args.steps, args.warmup, 1))
        state["weight"] = P.sample_weight(rank, world, new_spp) if split == "sample" and world > 1 else None
        options(False)
        tr.reserve(W, H, state["my_spp"])      # set-up, untimed like the upload: the timed region never allocates

    # The steps of a render are issued back to back with pipelining on: the tracer overlaps the latency-bound end of one
    # chunk of samples (deep bounce levels) with the start of the next on its own streams; rt_join orders the image
    # before the reduce.
    options(False)
    tr.set_pipelining(True)
    set_spp(spp)

    def trace_pass(i):
        tr.trace_device(W, H, 1 + i * state["my_spp"], state["my_spp"], img.data_ptr())

    def render(n):
        """One render: n passes accumulated on the device, the reduce onto rank 0 (N > 1), the mean copied to the host."""
        P.render(trace_pass, tr.join, img, n, split, world, weight=state["weight"])
        if rank == 0:
            host_img.copy_(img, non_blocking=True)

    grouped = dist.is_initialized()   # N > 1, or a single rank taken through the collectives (RT_FORCE_DIST=1)

    def timed_render(n):
        if grouped:
            dist.barrier()
        device_sync()
        t0 = time.perf_counter()
        render(n)
        if grouped:
            dist.barrier()
        device_sync()
        return time.perf_counter() - t0

    render(args.warmup)
    tr.sync()
    device_sync()
    # Samples per step: the workload's figure, raised (whole multiples, every rank alike) until the K timed steps last about
    # --min-seconds -- a timed region of under a second is within reach of clock ramps and of the driver's own sampling.
    if not args.spp and args.min_seconds > 0 and args.steps > 0:
        probe = timed_render(max(min(args.steps, 4), 1)) / max(min(args.steps, 4), 1)
        tr.sync()
        t = torch.tensor([probe], dtype=torch.float64, device=dev)
        if grouped:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        factor = int(min(max(-(-args.min_seconds // (float(t[0]) * args.steps)), 1), 64))
        if factor > 1:
            set_spp(state["spp"] * factor)
            render(min(args.warmup, 2))
            tr.sync()
            device_sync()
        note(f"a step of {spp} spp takes {float(t[0]) * 1e3:.2f} ms: {state['spp']} spp per step")
    spp, my_spp = state["spp"], state["my_spp"]
    my_pixels = W * H                                            # pixels this rank renders (tile split: its tiles')

    # work of one step: counted once, untimed, by the counting variant of the same kernels
    options(True)
    trace_pass(0)
    counts = tr.sync()
    # ... and, for flat scenes, by the counting variant of the walk that is timed (wide records, distance limit): scenes
    # beyond every cache are priced by the cheaper of the two walks' bytes
    walk_counts = None
    if wl.estimator == R.EST_SD_LOOP and wl.scene_bytes() > MALL_BYTES:
        options(2)
        trace_pass(0)
        walk_counts = tr.sync()
        assert walk_counts.rays == counts.rays and walk_counts.tri_hits == counts.tri_hits
    # the same rays walked in the reference's own visiting order over the reference's own boxes: what the reference
    # traversal (and the CPU baseline) has to do per ray -- on 16 spp (that walk is slow), skipped for scenes where it
    # would take minutes (the million-triangle fields)
    ref_counts = fast16 = None
    ref_spp = min(my_spp, 16)
    if rank == 0 and on_gpu and not args.no_reference_counts and len(wl.scene.nodes()) < 200000:
        options(True)
        tr.trace_device(W, H, 1, ref_spp, img.data_ptr())
        fast16 = tr.sync()
        options(True, R.TRAV_REFERENCE)
        tr.trace_device(W, H, 1, ref_spp, img.data_ptr())
        ref_counts = tr.sync()
        assert ref_counts.rays == fast16.rays and ref_counts.tri_hits == fast16.tri_hits
    options(False)
    note(f"scene uploaded, work counted ({counts.rays} rays per step)")

    render(min(args.warmup, 1))
    tr.sync()
    note("warm-up done")
    elapsed = timed_render(args.steps)
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
        alg_note = "16 per slab test on the grid records, 64 per leaf test"
        if walk_counts is not None:
            # the timed walk: a 64-byte record of eight child boxes per step (8 B per box tested), a 64-byte leaf record per
            # triangle test (its exact box is in the same record)
            wide_bytes = 8 * walk_counts.box_tests + 64 * walk_counts.tri_tests + 36 * walk_counts.tri_hits + image_bytes
            if wide_bytes < alg_bytes:
                alg_bytes = wide_bytes
                alg_note = ("8 per child box of a wide record, 64 per leaf test -- the walk with the distance limit, %.1f box tests per "
                            "ray against %.1f of the walk without" % (walk_counts.box_tests / max(walk_counts.rays, 1),
                                                                       counts.box_tests / max(counts.rays, 1)))
        alg_ops = traversal_ops(counts)
        secs = kernel_ms * 1e-3
        # hardware-counter figures of this workload, measured in separate rocprofv3 --pmc passes of one trace call and kept
        # with the profiles, together with a hash of the kernel sources (and of the library) they were measured on: figures
        # of other kernels than the ones loaded here are not reported.  Extensive figures (bytes) are scaled from the
        # measured call's samples per pixel to this step's.
        pmc, pmc_note = {}, "no counter figures for this workload under profiles/"
        stamp = source_stamp()
        ppath = os.path.join(ROOT, "profiles", "pmc_per_step.json")
        if os.path.exists(ppath) and on_gpu and world == 1:
            with open(ppath) as fh:
                pfile = json.load(fh)
            entry = pfile.get("workloads", {}).get(wl.name)
            measured_on = pfile.get("measured_on", {})
            if entry is not None and measured_on.get("kernel_sources_sha256") == stamp["kernel_sources_sha256"]:
                pmc = dict(entry)
                scale = my_spp / float(entry["spp"])
                for key in ("hbm_bytes_per_step", "hbm_read_bytes_per_step", "hbm_write_bytes_per_step"):
                    if pmc.get(key) is not None:
                        pmc[key] = int(pmc[key] * scale)
                pmc_note = "%s; bytes scaled from %d to %d samples per pixel" % (entry.get("source"), entry["spp"], my_spp)
            elif entry is not None:
                pmc_note = ("counter figures under profiles/ were measured on other kernel sources (%s) than the ones loaded (%s): dropped"
                            % (measured_on.get("kernel_sources_sha256"), stamp["kernel_sources_sha256"]))
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
                             % (alg_note if flat
                                else "32 per slab test, 36 per triangle test, 16 per sphere test", scene_bytes / 1e6))}
        roof["valu_frac"] = alg_ops / secs / 1e12 / VALU_PEAK_TOPS
        # what this GPU was measured to issue at best: 0.39 fp32 instructions per SIMD and clock with eight waves of independent
        # multiply-adds per SIMD, 0.33 with four (tools/valu_probe.hip, profiles/r03_valu_probe.txt) -- 61.5 T lane-operations/s
        roof["valu_peak_measured_Tops"] = 61.5
        roof.update({"kernel": "one step = one trace call, per chunk of 64 samples: flat scenes k_camera (scenes that cover little of the "
                               "frame) + one k_refill launch that follows every path to its end over the wide tree (small trees: "
                               "k_sliced first, for the fresh camera rays) + k_fold; pointer scenes k_path per bounce level, or "
                               "k_refill beyond the L2s, + k_fold",
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
                     "counters_from": pmc_note, "loaded": stamp})
        primary = my_pixels * my_spp if wl.bounce_limit > 0 or wl.estimator != R.EST_SD_LOOP else 0
        out = {
            "metric": "Mrays/s (all bounces)", "value": rays_total / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "msamples_per_s": (W * H * spp * args.steps) / elapsed / 1e6,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "Stanford Dragon mesh bundled with the reference (data/models), no external data"
                    if wl.estimator == R.EST_SD_LOOP else "synthetic",
            "config": {"workload": wl.name, "description": wl.description, "width": W, "height": H, "spp_per_step": spp,
                       "timed_region": "K steps accumulated into one running mean on the device" +
                                       (", one reduce of the image onto rank 0" if world > 1 else "") +
                                       ", the copy of the W*H*16-byte mean to page-locked host memory (rank 0); scene upload and BVH build excluded",
                       "split": split if world > 1 else None,
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
        if world == 1 and on_gpu and not args.no_extras:
            note("extra legs: camera rays alone, host-image boundary")
            # (1) camera rays alone: the same steps with bounce limit 1 (every path ends after its first search)
            options(False, bounce_limit=1)
            ksteps = max(min(args.steps, 8), 2)
            render(1)
            tr.sync()
            device_sync()
            t1 = time.perf_counter()
            render(ksteps)
            device_sync()
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
                               "note": "every step through the synchronous host-image boundary: rt_trace, one call per step, no overlap "
                                       "between steps, the W*H*16-byte mean copied to page-locked host memory inside each call "
                                       "(the reference's setData upload per frame, Texture.cpp:86-91)"}
        if world == 1 and on_gpu and not args.no_cpu_baseline:
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
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    tr.close()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
