"""Where one sketch pass spends its wall time, phase by phase (reset / push_device+sync / finish / copy-out), for a
given k and sketch size.  Debug aid for the large-n paths; not part of the bench contract.
usage (GPU box): python tools/phase_times.py --k 31 --n 2000000 [--gbases 10]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finch_rs_amd as F  # noqa: E402
import finch_rs_amd.sketch_schemes as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--n", type=int, default=2000000)
ap.add_argument("--gbases", type=float, default=10.0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--scale", type=float, default=0.0, help="> 0: a Scaled sketch (kmers_to_sketch = --n, this scale)")
a = ap.parse_args()
READ_LEN, GENOME_LEN, SEED = 150, 5_000_000, 20250620
n_reads = int(np.ceil(a.gbases * 1e9 / READ_LEN))
nbytes = n_reads * (READ_LEN + 1)
dg = F.DeviceBuffer(GENOME_LEN)
dr = F.DeviceBuffer(nbytes + 64)
S.synth_genome_device(dg, GENOME_LEN, SEED)
S.synth_reads_device(dr, dg, GENOME_LEN, 0, n_reads, READ_LEN, SEED, 10000, 500)
sk = (F.SketchParams.scaled(a.n, a.k, a.scale, 0) if a.scale > 0 else F.SketchParams.mash(a.n, a.n, True, a.k, 0)).create_sketcher()
for rep in range(a.reps):
    t = [time.perf_counter()]
    sk.reset(); sk.sync(); t.append(time.perf_counter())
    sk.push_device(dr.ptr, nbytes); sk.sync(); t.append(time.perf_counter())
    n, tk = sk.finish(); t.append(time.perf_counter())
    kc, km, pos = sk.to_arrays(); t.append(time.perf_counter())
    names = ["reset", "push_device+sync", "finish", "to_arrays"]
    print("rep %d: " % rep + "  ".join("%s %.2f ms" % (nm, (t[i + 1] - t[i]) * 1e3) for i, nm in enumerate(names)),
          " total %.2f ms  n=%d" % ((t[-1] - t[0]) * 1e3, n), flush=True)
