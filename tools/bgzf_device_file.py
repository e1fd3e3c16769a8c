"""Write one BGZF FASTQ (1.5 M reads of 150 bases; --noisy: random quality values) and sketch it twice through finch_sketch_files.
usage: python tools/bgzf_device_file.py [--noisy] [--level 1] [--reps 2]"""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
noisy = "--noisy" in sys.argv
level = int(sys.argv[sys.argv.index("--level") + 1]) if "--level" in sys.argv else 1
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
g = S.synth_genome_host(5_000_000, 7)
n_reads, rl = 1_500_000, 150
reads = S.synth_reads_host(g, 0, n_reads, rl, 7, 10000, 500).reshape(n_reads, rl + 1)[:, :rl]
q = np.random.default_rng(1).integers(35, 74, size=(n_reads, rl), dtype=np.uint8) if noisy else None
raw = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + (q[i].tobytes() if noisy else b"I" * rl) + b"\n" for i in range(n_reads))
path = "/tmp/bgzf_device_file.bgz"
with open(path, "wb") as f:
    for i in list(range(0, len(raw), 65280)) + [None]:
        ch = b"" if i is None else raw[i:i + 65280]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        c = co.compress(ch) + co.flush()
        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(c) + 25) + c + struct.pack("<II", zlib.crc32(ch), len(ch)))
p = S.SketchParams.mash(1000, 1000, True, 21, 0)
for rep in range(reps):
    t = time.time()
    H.sketch_files([path], p, H.FilterParams(False))
    dt = time.time() - t
    print("%.0f MB text as %.0f MB BGZF: %.3f s  %.2f Gbases/s" % (len(raw) / 1e6, os.path.getsize(path) / 1e6, dt, n_reads * rl / dt / 1e9), flush=True)
