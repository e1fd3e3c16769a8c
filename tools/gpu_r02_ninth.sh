#!/bin/bash
# full -m gpu suite, then the BGZF reader at several thread counts with FH_TRACE
export TMPDIR=/tmp
mkdir -p gpurun_out
nproc > gpurun_out/r02i_bgzf.txt
( time python -m pytest tests -m gpu -q -x --durations=8 ) > gpurun_out/r02i_pytest.log 2>&1
tail -15 gpurun_out/r02i_pytest.log
FH_TRACE=1 python - <<'PY' 2>&1 | grep -v "^\[fh\]" | tee -a gpurun_out/r02i_bgzf.txt
import os, sys, time, zlib, struct
sys.path.insert(0, ".")
import numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
g = S.synth_genome_host(5_000_000, 7)
n_reads, rl = 1_500_000, 150
reads = S.synth_reads_host(g, 0, n_reads, rl, 7, 10000, 500).reshape(n_reads, rl + 1)[:, :rl]
raw = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + b"I" * rl + b"\n" for i in range(n_reads))
def bgzf(data, block=65280):
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        ch = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(1, zlib.DEFLATED, -15); c = co.compress(ch) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(c) + 25) + c + struct.pack("<II", zlib.crc32(ch), len(ch)))
    return b"".join(out)
open("/tmp/t.bgz", "wb").write(bgzf(raw))
p = S.SketchParams.mash(1000, 1000, True, 21, 0)
for thr in (4, 8, 16, 32):
    os.environ["FINCH_READ_THREADS"] = str(thr)
    for rep in range(3):
        t = time.time(); H.sketch_files(["/tmp/t.bgz"], p, H.FilterParams(False)); dt = time.time() - t
        print("bgzf %d threads: %.3f s  %.2f GB/s text  %.2f Gbases/s" % (thr, dt, len(raw) / dt / 1e9, n_reads * rl / dt / 1e9))
PY
