#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r02h_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02h_bench.json').read())
print(d['value']/1e9)
for k,v in d['extras'].items(): print(k, {a:b for a,b in v.items() if a not in ('what','pmc')})
PY
FH_TRACE=1 FINCH_READ_THREADS=16 python - <<'PY' 2>&1 | grep -v "^\[fh\]" | tail -12
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
for rep in range(3):
    t = time.time(); H.sketch_files(["/tmp/t.bgz"], p, H.FilterParams(False)); dt = time.time() - t
    print("bgzf 16 threads: %.3f s  %.2f GB/s text" % (dt, len(raw) / dt / 1e9))
PY
