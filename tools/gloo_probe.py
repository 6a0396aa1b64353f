"""Two ranks sharing one GPU reduce a device tensor over gloo several times; tells whether a reduce that never returns
is the backend's doing or the tracer's, and which ingredient it needs.  Run under torch.distributed.run
--nproc-per-node 2 with a mode:
  plain        no tracer in the process
  barrier      no tracer; dist.barrier() + torch.cuda.synchronize() between the reduces (what bench.py does)
  tracer       the tracer renders into the tensor on torch's current stream (rt_set_stream) before each reduce
  tracer_sync  the same, torch.cuda.synchronize() + rt_sync before each reduce (nothing of the tracer in flight)
  tracer_own   the tracer keeps its own stream; rt_sync before each reduce
  tracer_idle  a tracer context exists (streams, events, workspace) but never renders
  busy         no tracer: about 50 ms of torch kernels writing the tensor are in flight on the current stream when the reduce is entered
  busy_side    no tracer: those kernels run on a side stream the current stream waits for (an event), as rt_join arranges it
  many         five overlapping 64-spp passes (rt_set_pipelining) in flight when the reduce is entered (bench.py's timed render)
  many_sync    the same, torch.cuda.synchronize() before the reduce
"""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(30, exit=True)
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
torch.cuda.set_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
W, H = 1920, 1080
t = torch.full((H, W, 4), float(rank + 1), device="cuda:0")
tr = None
if mode.startswith("tracer") or mode.startswith("many"):
    import raytracing_amd as R
    from raytracing_amd import workloads as WL
    tr = R.Tracer(0)
    tr.upload(R.SdScene().load_obj(WL.DRAGON_2426).finalize())
    tr.set_camera(WL.DRAGON_2426_CAMERA)
    tr.set_options(bounce_limit=4, tile_rank=rank, tile_world=2)
    if mode != "tracer_own":
        tr.set_stream(torch.cuda.current_stream().cuda_stream)
    if len(sys.argv) > 2 or mode.startswith("many"):
        tr.set_pipelining(True)
big = torch.ones((8192, 8192), device="cuda:0") if mode.startswith("busy") else None
side = torch.cuda.Stream() if mode == "busy_side" else None
for i in range(4):
    t0 = time.time()
    if big is not None:
        with torch.cuda.stream(side) if side is not None else torch.cuda.stream(torch.cuda.current_stream()):
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
            acc = big
            for k in range(40):
                acc = (acc @ big) * 1e-4
            t += acc[0, 0]
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
    if tr is not None and mode != "tracer_idle":
        t.zero_()
        if mode.startswith("many"):
            for k in range(5):
                tr.trace_device(W, H, 1 + 64 * k, 64, t.data_ptr())
        else:
            tr.trace_device(W, H, 1, 2, t.data_ptr())
        tr.join()
        if mode == "many_sync":
            torch.cuda.synchronize()
            print(f"[rank {rank}] GPU work of iteration {i} finished after {time.time() - t0:.3f} s", flush=True)
        if mode in ("tracer_sync", "tracer_own"):
            tr.sync()
            torch.cuda.synchronize()
    dist.reduce(t, dst=0)
    torch.cuda.synchronize()
    print(f"[rank {rank}] {mode} reduce {i} done in {time.time() - t0:.3f} s", flush=True)
    if mode != "plain":
        if tr is not None:
            tr.sync()
        dist.barrier()
        torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print(f"[rank {rank}] {mode} finished", flush=True)
