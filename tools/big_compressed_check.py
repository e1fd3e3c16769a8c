"""One-off check at a size the test suite does not reach: 1.3 GB of FASTQ text (several batches of the device-side BGZF
path, dozens of batches of the parallel gzip reader) as BGZF and as plain gzip must sketch like the plain file."""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
g = S.synth_genome_host(5_000_000, 11)
n_reads, rl = 4_200_000, 150
reads = S.synth_reads_host(g, 0, n_reads, rl, 11, 10000, 500).reshape(n_reads, rl + 1)[:, :rl]
rng = np.random.default_rng(2)
q = rng.integers(35, 74, size=(100_000, rl), dtype=np.uint8)
t0 = time.time()
raw = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + q[i % 100_000].tobytes() + b"\n" for i in range(n_reads))
print("text %.2f GB (%.0f s)" % (len(raw) / 1e9, time.time() - t0), flush=True)
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
plain, bg, gz = (os.path.join(d, "big_check." + e) for e in ("fastq", "bgz", "gz"))
open(plain, "wb").write(raw)
t0 = time.time()
with open(bg, "wb") as f:
    for i in list(range(0, len(raw), 65280)) + [None]:
        ch = b"" if i is None else raw[i:i + 65280]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        c = co.compress(ch) + co.flush()
        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(c) + 25) + c + struct.pack("<II", zlib.crc32(ch), len(ch)))
co = zlib.compressobj(1, zlib.DEFLATED, 31)
open(gz, "wb").write(co.compress(raw) + co.flush())
print("compressed (%.0f s): bgzf %.0f MB, gzip %.0f MB" % (time.time() - t0, os.path.getsize(bg) / 1e6, os.path.getsize(gz) / 1e6), flush=True)
del raw
p = S.SketchParams.mash(5000, 5000, True, 21, 0)
res = {}
for name, path in (("plain", plain), ("bgzf", bg), ("gzip", gz)):
    before = H.debug_device_inflate()
    t = time.time()
    r = H.sketch_files([path], p, H.FilterParams(False))
    dt = time.time() - t
    sk = r.sketch(0)
    res[name] = sk
    print("%s: %.3f s  %.2f Gbases/s  seq_length %d  device-inflated files +%d, re-read +%d" % (
        name, dt, n_reads * rl / dt / 1e9, sk.seq_length, H.debug_device_inflate()[0] - before[0], H.debug_device_inflate()[1] - before[1]), flush=True)
for name in ("bgzf", "gzip"):
    a, b = res["plain"], res[name]
    assert np.array_equal(a.arrays[0], b.arrays[0]) and np.array_equal(a.arrays[1], b.arrays[1]) and (a.seq_length, a.num_valid_kmers) == (b.seq_length, b.num_valid_kmers), name
print("same sketch from all three")
for f in (plain, bg, gz):
    os.remove(f)
