#!/bin/bash
# device-side BGZF inflate: its tests, then timings against the host-side inflate
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_bgzf_device.py -q -x --durations=5 ) > gpurun_out/r02j_pytest.log 2>&1
tail -30 gpurun_out/r02j_pytest.log
FH_TRACE=1 python - <<'PY' 2>&1 | grep -v "^\[fh\]" | tee gpurun_out/r02j_bgzf.txt
import os, sys, time, zlib, struct
sys.path.insert(0, ".")
import numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
g = S.synth_genome_host(5_000_000, 7)
n_reads, rl = 1_500_000, 150
reads = S.synth_reads_host(g, 0, n_reads, rl, 7, 10000, 500).reshape(n_reads, rl + 1)[:, :rl]
rng = np.random.default_rng(1)
def bgzf(data, level, block=65280):
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        ch = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15); c = co.compress(ch) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(c) + 25) + c + struct.pack("<II", zlib.crc32(ch), len(ch)))
    return b"".join(out)
p = S.SketchParams.mash(1000, 1000, True, 21, 0)
for name, qual in (("constant quality", None), ("noisy quality", 1)):
    if qual is None:
        raw = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + b"I" * rl + b"\n" for i in range(n_reads))
    else:
        q = rng.integers(35, 74, size=(n_reads, rl), dtype=np.uint8)
        raw = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n" for i in range(n_reads))
    for level in (1, 6):
        z = bgzf(raw, level)
        open("/tmp/t.bgz", "wb").write(z)
        for env in (None, "0"):
            if env is None: os.environ.pop("FINCH_DEVICE_INFLATE", None)
            else: os.environ["FINCH_DEVICE_INFLATE"] = env
            best = 1e9
            for rep in range(3):
                t = time.time(); H.sketch_files(["/tmp/t.bgz"], p, H.FilterParams(False)); best = min(best, time.time() - t)
            print("%s, level %d, %.0f MB -> %.0f MB, %s inflate: %.3f s  %.2f GB/s text  %.2f Gbases/s" % (name, level, len(raw) / 1e6, len(z) / 1e6, "device" if env is None else "host", best, len(raw) / best / 1e9, n_reads * rl / best / 1e9), flush=True)
PY
