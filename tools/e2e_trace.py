#!/usr/bin/env python3
"""Where an end-to-end call spends its time (GPU box): finch_sketch_buffer on a FASTQ text image in host memory with option trace,
by number of read threads.   python tools/e2e_trace.py [reads]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import finch_rs_amd as F
F.debug_set(trace="1")
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
RL, REC = 150, 151
dg = F.DeviceBuffer(5_000_000)
dr = F.DeviceBuffer(ns * REC + 64)
S.synth_genome_device(dg, 5_000_000, 20250620)
S.synth_reads_device(dr, dg, 5_000_000, 0, ns, RL, 20250620, 10000, 500)
reads = dr.download(ns * REC).reshape(ns, REC)[:, :RL]
w = 12 + RL + 3 + RL + 1
txt = np.empty((ns, w), np.uint8)
txt[:, 0], txt[:, 1] = ord("@"), ord("r")
idx = np.arange(ns, dtype=np.int64)
for d in range(9):
    txt[:, 10 - d] = 48 + (idx // 10 ** d) % 10
txt[:, 11] = 10
txt[:, 12:12 + RL] = reads
txt[:, 12 + RL:15 + RL] = np.frombuffer(b"\n+\n", np.uint8)
txt[:, 15 + RL:15 + 2 * RL] = ord("I")
txt[:, w - 1] = 10
data = txt.reshape(-1)
p = F.SketchParams.mash(1000, 1000, True, 21, 0)
for thr in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "4", "8", "16"]):
    F.debug_set(read_threads=thr)
    best = 1e30
    for it in range(3):
        print("---- read threads %s, pass %d" % (thr, it), file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        res = H.sketch_stream(data, "fastq", p, H.FilterParams(False), device=0)
        best = min(best, time.perf_counter() - t0)
    print("read threads %2s: %.1f ms  %.2f GB/s of text  %.2f Gbases/s" % (thr, best * 1e3, data.size / best / 1e9, ns * RL / best / 1e9), flush=True)
