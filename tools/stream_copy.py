#!/usr/bin/env python3
"""HBM stream-copy ceiling of the box (what a plain device-to-device copy reaches), to put next to the 8 TB/s nominal peak
the roofline fields are priced against.  Prints one JSON line."""
import json
import time

import torch

n = 1 << 30  # 4 GiB of float32 read + 4 GiB written per copy
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    b.copy_(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"what": "torch device-to-device copy, 4 GiB read + 4 GiB written", "ms": dt * 1e3, "GB_per_s": 2 * 4 * n / dt / 1e9,
                  "device": torch.cuda.get_device_name(0)}))
