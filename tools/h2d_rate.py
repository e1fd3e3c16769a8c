#!/usr/bin/env python3
"""Pinned host-to-device copy rate of this box's PCIe link (what binds configs[4] once the kernels are out of the way):
one stream, then 2 / 4 / 16 streams at once, copy sizes 4 / 16 / 64 MiB.  usage: python tools/h2d_rate.py"""
import time
import torch

dev = torch.device("cuda:0")
for mib in (4, 16, 64):
    n = mib << 20
    for ns in (1, 2, 4, 16):
        hs = [torch.empty(n, dtype=torch.uint8, pin_memory=True) for _ in range(ns)]
        ds = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(ns)]
        st = [torch.cuda.Stream(dev) for _ in range(ns)]
        reps = max(4, 2048 // (mib * ns))
        for w in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for r in range(reps):
                for i in range(ns):
                    with torch.cuda.stream(st[i]):
                        ds[i].copy_(hs[i], non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("H2D pinned %3d MiB x %2d streams: %6.1f GB/s" % (mib, ns, reps * ns * n / dt / 1e9), flush=True)
# and the other way, for the results
h = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True)
d = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for r in range(16):
    h.copy_(d, non_blocking=True)
torch.cuda.synchronize()
print("D2H pinned  64 MiB x  1 stream : %6.1f GB/s" % (16 * (64 << 20) / (time.perf_counter() - t0) / 1e9))
# host memcpy rate of one core (what a worker's strip pass is held against)
import numpy as np
a = np.frombuffer(bytearray(256 << 20), dtype=np.uint8); b = np.empty_like(a)
t0 = time.perf_counter(); b[:] = a; b[:] = a
print("host memcpy, one thread: %.1f GB/s" % (2 * a.nbytes / (time.perf_counter() - t0) / 1e9))
