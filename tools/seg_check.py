#!/usr/bin/env python3
"""Does a block of synthetic reads go through the segment kernel, and how fast?  python tools/seg_check.py [gbases] [k] [n]   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
k = int(sys.argv[2]) if len(sys.argv) > 2 else 21
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
RL, GL, SEED = 150, 5_000_000, 20250620
nr = int(gb * 1e9 / RL)
dg = F.DeviceBuffer(GL); dr = F.DeviceBuffer(nr * (RL + 1) + 64)
S.synth_genome_device(dg, GL, SEED); S.synth_reads_device(dr, dg, GL, 0, nr, RL, SEED, 10000, 500)
for hint in (0, 151, 1):
    sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher()
    sk.set_profiling(True)
    sk.set_record_stride(hint)
    best = 1e9
    for it in range(4):
        t0 = time.perf_counter(); sk.reset(); sk.push_device(dr.ptr, nr * (RL + 1)); kc, km, _ = sk.to_arrays(); tk = sk.finish()[1]; dt = time.perf_counter() - t0
        ms, nl, npos = sk.kernel_time()
        if it: best = min(best, dt)
    print("hint %3d: %.3f ms/pass  %.1f Gbases/s  kernel %.3f ms  segments(launches, probes, stride) %s  xor %x kmers %d" %
          (hint, best * 1e3, nr * RL / best / 1e9, ms, sk.debug_segments(), int(np.bitwise_xor.reduce(kc["hash"])), tk), flush=True)
    sk.close()
