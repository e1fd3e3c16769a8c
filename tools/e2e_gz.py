"""End-to-end rate from a gzip-compressed FASTQ (inflate + parse on the host, one thread per file).
usage (GPU box): python tools/e2e_gz.py"""
import gzip, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F  # noqa: E402
g = S.synth_genome_host(5_000_000, 7)
n_reads, rl = 1_500_000, 150
reads = S.synth_reads_host(g, 0, n_reads, rl, 7, 10000, 500).reshape(n_reads, rl + 1)[:, :rl]
qual = b"I" * rl
raw = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + qual + b"\n" for i in range(n_reads))
path = "/tmp/e2e.fastq.gz"
t = time.time()
with gzip.open(path, "wb", compresslevel=1) as f:
    f.write(raw)
print("wrote %.0f MB text as %.0f MB gz in %.1f s" % (len(raw) / 1e6, os.path.getsize(path) / 1e6, time.time() - t))
p = S.SketchParams.mash(1000, 1000, True, 21, 0)
for rep in range(3):
    t = time.time(); res = H.sketch_files([path], p, H.FilterParams(False)); dt = time.time() - t
    print("sketch_files fastq.gz: %.2f s  %.2f GB/s of inflated text  %.1f Mbases/s" % (dt, len(raw) / dt / 1e9, n_reads * rl / dt / 1e6))
import zlib
t = time.time(); d = zlib.decompressobj(31); n = 0
with open(path, "rb") as f:
    while True:
        b = f.read(1 << 20)
        if not b: break
        n += len(d.decompress(b))
print("zlib inflate alone (python): %.2f s  %.2f GB/s" % (time.time() - t, n / (time.time() - t) / 1e9))
# the same text as BGZF (bgzip): independent 64 KiB members, inflated by FINCH_READ_THREADS threads
import struct
def bgzf(data, block=65280):
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        ch = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(1, zlib.DEFLATED, -15); c = co.compress(ch) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(c) + 25) + c + struct.pack("<II", zlib.crc32(ch), len(ch)))
    return b"".join(out)
bpath = "/tmp/e2e.fastq.bgz"
t = time.time()
with open(bpath, "wb") as f: f.write(bgzf(raw))
print("wrote the same text as %.0f MB BGZF in %.1f s" % (os.path.getsize(bpath) / 1e6, time.time() - t))
ref = res.sketch(0).arrays[0]
for thr in ("1", "2", "4", "8", "16"):
    F.debug_set(read_threads=thr)
    best = 1e9
    for rep in range(2):
        t = time.time(); rb = H.sketch_files([bpath], p, H.FilterParams(False)); best = min(best, time.time() - t)
    assert np.array_equal(rb.sketch(0).arrays[0], ref)
    print("sketch_files fastq BGZF, %2s threads: %.2f s  %.2f GB/s of inflated text  %.1f Mbases/s" % (thr, best, len(raw) / best / 1e9, n_reads * rl / best / 1e6))
