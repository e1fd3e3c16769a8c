#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
FH_TRACE=1 FINCH_READ_THREADS=16 timeout 600 python - <<'PY' 2>&1 | grep -v "^\[fh\]" | tail -30
import os, sys, time, zlib
sys.path.insert(0, ".")
import numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
g = S.synth_genome_host(5_000_000, 7)
n_reads, rl = 1_500_000, 150
reads = S.synth_reads_host(g, 0, n_reads, rl, 7, 10000, 500).reshape(n_reads, rl + 1)[:, :rl]
raw = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + b"I" * rl + b"\n" for i in range(n_reads))
co = zlib.compressobj(6, zlib.DEFLATED, 31)
z = co.compress(raw) + co.flush()
open("/tmp/x.fastq.gz", "wb").write(z)
p = S.SketchParams.mash(1000, 1000, True, 21, 0)
for rep in range(3):
    t = time.time(); H.sketch_files(["/tmp/x.fastq.gz"], p, H.FilterParams(False)); print("sketch_files %.3f s" % (time.time() - t), flush=True)
for rep in range(3):
    t = time.time(); got = H.source_probe(z, 64 << 20, len(raw) + 4096); dt = time.time() - t
    print("source_probe (no device) %.3f s" % dt, got == raw, flush=True)
PY
