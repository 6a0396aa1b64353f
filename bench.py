#!/usr/bin/env python3
"""Throughput benchmark of the HIP path tracer: Mrays/s (all bounces), one JSON line on rank 0.

A step is one trace pass (spp samples per pixel) over the whole frame of the workload.  With N > 1
(launched under torch.distributed.run, one rank per GPU) every rank traces the same frame with its own
sample-index range (weak scaling: spp per GPU fixed) and the per-rank means are summed onto rank 0 with
one RCCL reduce per step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec
VALU_PEAK_TOPS = 78.6         # 256 CU x 128 lanes x 2.4 GHz, one op per lane-cycle (the path forbids FMA)


def cpu_baseline(wl, args):
    """Times the CPU oracle (a port of the reference path) on a bounded sample of the workload: the same
    scene, camera and bounce limit at reduced resolution and a few spp (about 10-20 s of CPU work)."""
    from oracle import objparse, oracle as O
    from raytracing_amd import workloads as WL
    import raytracing_amd as R
    threads = min(os.cpu_count() or 1, 16)
    div = args.cpu_div
    w, h = max(wl.width // div, 16), max(wl.height // div, 9)
    cam = wl.camera
    ocam = O.make_camera(cam.focal_length, cam.position, cam.look_at, cam.width, cam.aspect_ratio)
    if wl.estimator == R.EST_SD_LOOP:
        osc = O.SdScene()
        tris, sizes = objparse.load_obj(wl.obj_path)
        if wl.field_k:
            tris = WL.dragon_grid_tris(tris, wl.field_k)
            sizes = [len(tris)]
        osc.add_model(tris, sizes)
        osc.finalize()
    else:
        osc = O.OldScene()
        for c, rad, kind, alb, param in (WL.rtiow_spheres() if wl.name.startswith("rtiow") else WL.small_sphere_scene()):
            osc.add_sphere(c, rad, kind, alb, param)
        osc.build()
    # (ii) fair host schedule: one row band per core, no per-pixel lock
    # (polynomial cos/sin so that its visit counts can be compared with the GPU's; the cost of the two
    # trig evaluations per bounce is negligible next to the traversal)
    opt = O.make_options(bounce_limit=wl.bounce_limit, seed=wl.seed, jitter=wl.jitter, trig=O.TRIG_POLY,
                         threads=threads, lock_per_pixel=0, accum=O.ACCUM_FORWARD)
    # the sample is sized from a 1-spp probe so that each of the two timed renders is about args.cpu_seconds of
    # CPU work whatever the workload costs per ray
    _, _, probe = osc.render(ocam, opt, w, h, 1, 1)
    spp = args.cpu_spp or int(min(max(args.cpu_seconds / max(probe.seconds, 1e-4), 1), 256))
    args.cpu_spp = spp
    _, _, st = osc.render(ocam, opt, w, h, 1, spp)
    # (i) the reference's own schedule: 16 row bands per sample and a shared_lock per pixel
    opt_ref = O.make_options(bounce_limit=wl.bounce_limit, seed=wl.seed, jitter=wl.jitter, trig=O.TRIG_LIBM,
                             threads=16, lock_per_pixel=1, accum=O.ACCUM_RECURSIVE)
    _, _, st_ref = osc.render(ocam, opt_ref, w, h, 1, spp)
    return {"value": st.rays / st.seconds / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{wl.name} scene/camera/bounce limit at {w}x{h}, {args.cpu_spp} spp ({st.rays} rays, {st.seconds:.2f} s)",
            "reference_schedule_value": st_ref.rays / st_ref.seconds / 1e6,
            "reference_schedule": "16 std::async row bands + shared_lock per pixel"}, st, (w, h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dragon18520_1080p")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per step (0 = the workload's)")
    ap.add_argument("--split", choices=("sample", "tile"), default="sample",
                    help="N > 1: sample = every rank renders its own samples of the whole frame (weak scaling, the default); "
                         "tile = the ranks share the tiles of one frame (strong scaling)")
    ap.add_argument("--traversal", type=int, default=4, help="RT_TRAV_* of the timed steps (4 = fast walk, 7 = with lane re-fill)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-counts", action="store_true",
                    help="skip the untimed pass that counts the tests of the reference-order walk")
    ap.add_argument("--cpu-div", type=int, default=4, help="cpu_baseline sample: resolution divisor")
    ap.add_argument("--cpu-spp", type=int, default=0, help="cpu_baseline sample: samples per pixel (0 = sized from a probe)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="cpu_baseline sample: target seconds per timed render")
    args = ap.parse_args()
    if os.environ.get("RT_BENCH_WATCHDOG"):  # dump every thread's stack if the run is still going after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RT_BENCH_WATCHDOG"]), exit=True)

    import numpy as np
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
    spp = args.spp or wl.spp
    W, H = wl.width, wl.height
    tr = R.Tracer(local_rank)
    tr.upload(wl.scene)
    tr.set_camera(wl.camera)
    tr.set_stream(torch.cuda.current_stream().cuda_stream)
    img = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)

    tile_split = args.split == "tile" and world > 1

    def options(stats, traversal=args.traversal):
        if tile_split:  # the same samples on every rank, disjoint 8x8 tiles (tile k belongs to rank k % world)
            tr.set_options(bounce_limit=wl.bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter,
                           collect_stats=stats, tile_rank=rank, tile_world=world, traversal=traversal)
        else:
            tr.set_options(bounce_limit=wl.bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter,
                           collect_stats=stats, sample_offset=P.sample_offset(rank, spp * max(args.steps, args.warmup)),
                           traversal=traversal)

    # One render = `n` steps of `spp` samples accumulated into the running mean on the device (the reference's
    # progressive sampleCount protocol, Renderer/TracerImpl.cpp:9-12), then -- N > 1 -- ONE RCCL reduce(SUM) of the
    # per-rank means, / world on rank 0.  Rank r draws the sample indices r * n * spp + 1 ... (r + 1) * n * spp.
    # The steps are issued back to back with pipelining on: the tracer overlaps the latency-bound end of one step (deep
    # bounce levels) with the start of the next on its own streams; rt_join orders the image before the reduce.
    tr.set_pipelining(True)

    def render(n):
        for i in range(n):
            tr.trace_device(W, H, 1 + i * spp, spp, img.data_ptr())
        tr.join()
        if tile_split:  # pixels of other ranks are zero here: one reduce(SUM) assembles the frame exactly
            P.combine_tile_split(img, dst=0)
        else:
            P.combine_sample_split(img, dst=0)

    # work of one trace call: counted once, untimed, by the counting variant of the same kernel
    options(True)
    tr.trace_device(W, H, 1, spp, img.data_ptr())
    counts = tr.sync()
    # the same rays walked in the reference's own visiting order over the reference's own boxes: what the
    # reference traversal (and the CPU baseline) has to do for this frame
    ref_counts = None
    # (skipped for scenes where that walk would take minutes: the million-triangle fields)
    if rank == 0 and not args.no_reference_counts and len(wl.scene.nodes()) < 200000:
        options(True, R.TRAV_REFERENCE)
        tr.trace_device(W, H, 1, spp, img.data_ptr())
        ref_counts = tr.sync()
        assert ref_counts.rays == counts.rays and ref_counts.tri_hits == counts.tri_hits
    options(False)
    note(f"scene uploaded, work counted ({counts.rays} rays per step)")

    # set-up, untimed like the upload: both of the tracer's internal pipes get their workspace (a step allocates it on
    # first use), so that the timed region never allocates whatever W is
    for i in range(2):
        tr.trace_device(W, H, 1 + i * spp, spp, img.data_ptr())
    tr.sync()
    render(args.warmup)
    tr.sync()
    note("warm-up done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    render(args.steps)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    note(f"{args.steps} timed steps in {elapsed * 1e3:.2f} ms")
    st = tr.sync()
    assert st.launches == args.steps  # st.rays: closest-hit searches of all the timed steps of this rank

    total = torch.tensor([elapsed, float(st.rays)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = total.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        elapsed, rays_total = float(tmax[0]), float(total[1])
    else:
        rays_total = float(total[1])

    if rank == 0:
        kernel_ms = st.kernel_ms_sum / st.launches

        def traversal_bytes(c):
            # 32 B per slab test (box + 2 links), 36 B per triangle test (3 positions), 16 B per sphere test,
            # 36 B of normals per traversal that ends on a triangle
            return 32 * c.box_tests + 36 * c.tri_tests + 16 * c.sph_tests + 36 * c.tri_hits

        def traversal_ops(c):
            return 27 * c.box_tests + 56 * c.tri_tests + 24 * c.sph_tests + 60 * c.rays

        # per pixel sample: one 16 B colour written and read once (the fold); per pixel: the 16 B mean written
        image_bytes = (32 * W * H * spp + 16 * W * H) // (world if tile_split else 1)
        alg_bytes = traversal_bytes(counts) + image_bytes
        alg_ops = traversal_ops(counts)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh).get(f"{wl.name}:{spp}:fast")
        out = {
            "metric": "Mrays/s (all bounces)", "value": rays_total / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "msamples_per_s": (W * H * spp * args.steps * (1 if tile_split else world)) / elapsed / 1e6,
            "higher_is_better": True, "scaling": "strong" if tile_split else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "Stanford Dragon mesh bundled with the reference (data/models), no external data"
                    if wl.estimator == R.EST_SD_LOOP else "synthetic",
            "config": {"workload": wl.name, "width": W, "height": H, "spp_per_step_per_gpu": spp,
                       "bounce_limit": wl.bounce_limit, "rays_per_step_per_gpu": int(st.rays // args.steps),
                       "traversal": "fast walk (own hierarchy over the reference's leaf boxes; hits bit-identical "
                                    "to the reference-order walk), primitive tests noted and run 64 at a time",
                       "parallelism": "single GPU" if world == 1 else
                                      "scene replicated, interleaved 8x8 image tiles per GPU (tile k -> GPU k % N), steps accumulate on "
                                      "the device, one RCCL reduce of the zero-padded W*H*4 float image at the end of the render"
                                      if tile_split else
                                      "scene replicated, sample-index range per GPU, steps accumulate on the device, one RCCL "
                                      "reduce of the W*H*4 float mean at the end of the render"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_sliced (flat scenes within the L2s; k_path for pointer scenes, k_refill beyond the L2s), one launch per round / bounce level, + k_fold = one trace call",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": int(alg_bytes),
                         "note": "algorithmic bytes = the tests this walk performs x their record payload + the "
                                 "per-sample colour traffic; node and triangle records are L2-resident for this "
                                 "scene, so most of these bytes never reach HBM (see traffic)"},
            "valu": {"achieved": alg_ops / (kernel_ms * 1e-3) / 1e12, "peak": VALU_PEAK_TOPS, "unit": "Tops/s",
                     "frac": alg_ops / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                     "algorithmic_ops_per_launch": int(alg_ops)},
            "counts_per_launch": {"rays": int(counts.rays), "box_tests": int(counts.box_tests),
                                  "tri_tests": int(counts.tri_tests), "sph_tests": int(counts.sph_tests),
                                  "tri_hits": int(counts.tri_hits), "replays": int(counts.replays)},
        }
        if ref_counts is not None:
            rb = traversal_bytes(ref_counts) + 16 * W * H
            out["reference_order_equivalent"] = {
                "box_tests": int(ref_counts.box_tests), "tri_tests": int(ref_counts.tri_tests),
                "sph_tests": int(ref_counts.sph_tests), "algorithmic_bytes_per_launch": int(rb),
                "equivalent_GBps": rb / (kernel_ms * 1e-3) / 1e9,
                "note": "tests and bytes the reference's own traversal needs for the same rays and the same hits"}
        if world == 1 and not args.no_cpu_baseline:
            cb, ost, (cw, ch) = cpu_baseline(wl, args)
            # same bounded sample on the GPU: counts must equal the oracle's (checker, not the thing measured)
            tr.set_options(bounce_limit=wl.bounce_limit, estimator=wl.estimator, seed=wl.seed, jitter=wl.jitter,
                           collect_stats=True, traversal=R.TRAV_REFERENCE)
            _, _, gst = tr.trace(cw, ch, 1, args.cpu_spp, want_image=False)
            cb["gpu_counts_match_oracle"] = bool((gst.rays, gst.box_tests, gst.tri_tests, gst.sph_tests) ==
                                                 (ost.rays, ost.box_tests, ost.tri_tests, ost.sph_tests))
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    tr.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
