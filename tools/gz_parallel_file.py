"""Write one gzip FASTQ (1.5 M reads of 150 bases; --noisy: random quality values) and sketch it through finch_sketch_files
with several values of FINCH_READ_THREADS.  usage: python tools/gz_parallel_file.py [--noisy] [--level 1]"""
import gzip, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F  # noqa: E402
noisy = "--noisy" in sys.argv
level = int(sys.argv[sys.argv.index("--level") + 1]) if "--level" in sys.argv else 1
g = S.synth_genome_host(5_000_000, 7)
n_reads, rl = 1_500_000, 150
reads = S.synth_reads_host(g, 0, n_reads, rl, 7, 10000, 500).reshape(n_reads, rl + 1)[:, :rl]
q = np.random.default_rng(1).integers(35, 74, size=(n_reads, rl), dtype=np.uint8) if noisy else None
raw = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + (q[i].tobytes() if noisy else b"I" * rl) + b"\n" for i in range(n_reads))
path = "/tmp/gz_parallel_file.fastq.gz"
co = zlib.compressobj(level, zlib.DEFLATED, 31)
with open(path, "wb") as f:
    f.write(co.compress(raw) + co.flush())
p = S.SketchParams.mash(1000, 1000, True, 21, 0)
print("%.0f MB text as %.0f MB gzip (level %d, %s quality)" % (len(raw) / 1e6, os.path.getsize(path) / 1e6, level, "random" if noisy else "constant"), flush=True)
for thr in (1, 4, 8, 16, 32):
    F.debug_set(read_threads=str(thr))
    best = 1e9
    for rep in range(3):
        t = time.time()
        H.sketch_files([path], p, H.FilterParams(False))
        best = min(best, time.time() - t)
    print("  %2d threads: %.3f s  %.2f GB/s text  %.2f Gbases/s" % (thr, best, len(raw) / best / 1e9, n_reads * rl / best / 1e9), flush=True)
