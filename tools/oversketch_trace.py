#!/usr/bin/env python3
"""What one pass over configs[1]'s resident stream is made of at a given (k, n): the launches the option trace logs, the sketch kernel's own
time, the pass's wall time.   FH_DEBUG=trace python tools/oversketch_trace.py [k [n [gbases]]]      (on an MI355X; under
`rocprofv3 --kernel-trace --stats` it also gives the per-kernel totals)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import finch_rs_amd as F  # noqa: E402
from finch_rs_amd import sketch_schemes as S  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
gb = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
RL, REC, GL, SEED = 150, 151, 5_000_000, 20250620
n_reads = int(np.ceil(gb * 1e9 / RL))
dg = F.DeviceBuffer(GL)
dr = F.DeviceBuffer(n_reads * REC + 64)
S.synth_genome_device(dg, GL, SEED)
S.synth_reads_device(dr, dg, GL, 0, n_reads, RL, SEED, 10000, 500)
s = F.SketchParams.mash(n, n, True, k, 0).create_sketcher()
s.set_profiling(True)
for it in range(int(os.environ.get("PASSES", "3"))):
    sys.stderr.write("== pass %d\n" % it)
    t0 = time.perf_counter()
    s.reset()
    s.push_device(dr.ptr, n_reads * REC)
    nn, tk = s.finish()
    dt = time.perf_counter() - t0
    ms, nl, npos = s.kernel_time()
    print("pass %d: %.3f ms wall, sketch kernels %.3f ms in %d launches over %.3f G positions (%.1f GB/s), %d hashes, segments %s"
          % (it, dt * 1e3, ms, nl, npos / 1e9, npos / 1e9 / (ms / 1e3) if ms else 0, nn, s.debug_segments()), flush=True)
