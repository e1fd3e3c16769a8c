"""End-to-end (file on disk -> sketch) rates of the host layer, PCIe and parsing included.  Not the bench metric."""
import os, sys, time, gzip
sys.path.insert(0, ".")
import numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F  # noqa: E402
from finch_rs_amd.sketch_schemes import SketchParams

n_reads, rl = 4_000_000, 150
g = S.synth_genome_host(5_000_000, 1)
t = time.time()
reads = S.synth_reads_host(g, 0, n_reads, rl, 1, 10000, 500).reshape(n_reads, rl + 1)[:, :rl]
qual = b"I" * rl
path = "/tmp/e2e.fastq"
with open(path, "wb") as f:
    for i in range(n_reads):
        f.write(b"@r%d\n" % i); f.write(reads[i].tobytes()); f.write(b"\n+\n"); f.write(qual); f.write(b"\n")
size = os.path.getsize(path)
print("wrote %.2f GB fastq in %.1fs" % (size / 1e9, time.time() - t))
p = SketchParams.mash(1000, 1000, True, 21, 0)
F.debug_set(device_parse="0")
for rep in range(2):
    t = time.time(); res = H.sketch_files([path], p, H.FilterParams(False)); dt = time.time() - t
    print("sketch_files fastq, host parser: %.2f s  %.2f GB/s text  %.1f Mbases/s" % (dt, size / dt / 1e9, n_reads * rl / dt / 1e6))
F.debug_set(device_parse="1")
for rep in range(2):
    t = time.time(); res2 = H.sketch_files([path], p, H.FilterParams(False)); dt = time.time() - t
    print("sketch_files fastq, device-side parsing: %.2f s  %.2f GB/s text  %.1f Mbases/s" % (dt, size / dt / 1e9, n_reads * rl / dt / 1e6))
assert np.array_equal(res.sketch(0).arrays[0], res2.sketch(0).arrays[0])
F.debug_set(device_parse=None)
# FASTA genome-like
fa = "/tmp/e2e.fa"
seq = S.synth_genome_host(200_000_000, 7).tobytes()
with open(fa, "wb") as f:
    f.write(b">chr\n")
    for i in range(0, len(seq), 70 * 100000):
        blk = seq[i:i + 70 * 100000]
        f.write(b"\n".join(blk[j:j + 70] for j in range(0, len(blk), 70))); f.write(b"\n")
for dev in (0, 0, 1, 1):
    F.debug_set(device_parse="1" if dev else "0")
    t = time.time(); r = H.sketch_files([fa], p, H.FilterParams(False)); dt = time.time() - t
    print("sketch_files fasta 200 Mb%s: %.2f s  %.1f Mbases/s" % (", device-side parsing" if dev else "", dt, 200e6 / dt / 1e6))
    if not dev:
        res = r
assert np.array_equal(res.sketch(0).arrays[0], r.sketch(0).arrays[0]) and res.sketch(0).seq_length == r.sketch(0).seq_length
F.debug_set(device_parse=None)
# batch of small fastas across threads
paths = []
for i in range(256):
    pth = "/tmp/e2e_%d.fa" % i
    with open(pth, "wb") as f:
        f.write(b">g\n"); s5 = seq[(i % 60) * 3_000_000:(i % 60 + 1) * 3_000_000 + 2_000_000]
        f.write(b"\n".join(s5[j:j + 70] for j in range(0, len(s5), 70))); f.write(b"\n")
    paths.append(pth)
for dev in (0, 1):
    F.debug_set(device_parse="1" if dev else "0")
    for nt in (1, 4, 8, 16):
        t = time.time(); res = H.sketch_files(paths, p, H.FilterParams(False), n_threads=nt); dt = time.time() - t
        print("batch 256 x 5 Mb fasta%s, %d threads: %.2f s  %.1f files/s  %.1f Mbases/s"
              % (", device-side parsing" if dev else "", nt, dt, 256 / dt, 256 * 5e6 / dt / 1e6))
