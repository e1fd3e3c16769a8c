import sys, time
sys.path.insert(0, ".")
import numpy as np
import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
n_reads = 20_000_000; rl = 150; rec = rl + 1
dg = F.DeviceBuffer(5_000_000); dr = F.DeviceBuffer(n_reads * rec + 64)
S.synth_genome_device(dg, 5_000_000, 1); S.synth_reads_device(dr, dg, 5_000_000, 0, n_reads, rl, 1, 10000, 500)
sk = F.SketchParams.mash(2_000_000, 2_000_000, True, 31, 0).create_sketcher()
for rep in range(3):
    t0 = time.perf_counter(); sk.reset(); t1 = time.perf_counter()
    sk.push_device(dr.ptr, n_reads * rec); sk.sync(); t2 = time.perf_counter()
    n, tk = sk.finish(); t3 = time.perf_counter()
    kc, km, pos = sk.to_arrays(); t4 = time.perf_counter()
    print("reset %.1f ms  push+sync %.1f ms  finish %.1f ms  to_arrays %.1f ms  counters %s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, sk.debug_counters()))
