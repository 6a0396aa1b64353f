"""One process, no collective: what a collective backend does with a device tensor the tracer is still writing --
an event recorded on torch's current stream, a side stream that waits for it and copies the tensor to pinned host
memory, a host wait for the side stream (from a worker thread, as gloo does, or from the main thread).
mode: null (torch's current stream is the legacy default stream) | side (a torch.cuda.Stream is current)
      +thread (the host wait happens on another thread)."""
import faulthandler
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(30, exit=True)
import raytracing_amd as R
from raytracing_amd import workloads as WL

mode = sys.argv[1] if len(sys.argv) > 1 else "null"
W, H = 1920, 1080
torch.cuda.set_device(0)
cur = torch.cuda.Stream() if mode.startswith("side") else torch.cuda.current_stream()
with torch.cuda.stream(cur):
    t = torch.zeros((H, W, 4), device="cuda:0")
    pinned = torch.empty((H, W, 4), pin_memory=True)
    tr = R.Tracer(0)
    tr.upload(R.SdScene().load_obj(WL.DRAGON_2426).finalize())
    tr.set_camera(WL.DRAGON_2426_CAMERA)
    tr.set_options(bounce_limit=4)
    tr.set_stream(torch.cuda.current_stream().cuda_stream)
    tr.set_pipelining(True)
    side = torch.cuda.Stream()
    for i in range(3):
        t0 = time.time()
        for k in range(5):
            tr.trace_device(W, H, 1 + 64 * k, 64, t.data_ptr())
        tr.join()
        ev = torch.cuda.Event()
        ev.record()                      # on the current stream, behind the join
        side.wait_event(ev)
        with torch.cuda.stream(side):
            pinned.copy_(t, non_blocking=True)
        if mode.endswith("thread"):
            th = threading.Thread(target=side.synchronize)
            th.start()
            th.join()
        else:
            side.synchronize()
        print(f"{mode}: iteration {i} copied after {time.time() - t0:.3f} s, alpha mean {float(pinned[..., 3].mean()):.4f}", flush=True)
        tr.sync()
    tr.close()
print(f"{mode}: finished", flush=True)
