#!/usr/bin/env python3
"""A batch of synthetic FASTA genomes (bench.py's configs[4] generator, files 0..N-1) through ONE finch_sketch_files call, for
profiling: `rocprofv3 --kernel-trace --stats -- python tools/batch_trace.py [files] [threads]` shows what the GPU does per file.
Prints files/s and Gbases/s of the timed call."""
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finch_rs_amd import host as H, sketch_schemes as S  # noqa: E402
import finch_rs_amd as F  # noqa: E402

SEED = 20250620


def _write(job):
    d, i = job
    with open(os.path.join(d, "g%05d.fa" % i), "wb") as f:
        f.write(S.synth_fasta_file(i, SEED))
    return S.synth_fasta_length(i, SEED)


nf = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d = tempfile.mkdtemp(prefix="finch_bt_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
        lens = pool.map(_write, [(d, i) for i in range(nf)], chunksize=8)
    paths = [os.path.join(d, "g%05d.fa" % i) for i in range(nf)]
    # warm: handles, page cache, and the staging buffers grown to the largest file (a slot pins only what its pushes needed)
    H.sketch_files(paths if len(sys.argv) <= 3 else paths[:48], F.SketchParams.default(), H.FilterParams(None), n_threads=nt)
    t0 = time.perf_counter()
    res = H.sketch_files(paths, F.SketchParams.default(), H.FilterParams(None), n_threads=nt)
    dt = time.perf_counter() - t0
    print("%d files, %d threads (0 = default): %.3f s  %.0f files/s  %.2f Gbases/s" % (nf, nt, dt, nf / dt, sum(lens) / dt / 1e9))
finally:
    shutil.rmtree(d, ignore_errors=True)
