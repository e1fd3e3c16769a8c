"""A batch of 5 Mb FASTA genomes with the CLI's default parameters (kmers_to_sketch = 1000 x oversketch 200 = 200 000,
final size 1000) next to the plain n = 1000 sketch.  usage (GPU box): python tools/batch_oversketch.py [n_files]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finch_rs_amd import host as H, sketch_schemes as S
n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 512
seq = S.synth_genome_host(200_000_000, 7).tobytes()
distinct = []
for i in range(256):
    pth = "/tmp/e2e_%d.fa" % i
    if not os.path.exists(pth):
        with open(pth, "wb") as f:
            f.write(b">g\n"); s5 = seq[(i % 60) * 3_000_000:(i % 60 + 1) * 3_000_000 + 2_000_000]
            f.write(b"\n".join(s5[j:j + 70] for j in range(0, len(s5), 70))); f.write(b"\n")
    distinct.append(pth)
paths = [distinct[i % 256] for i in range(n_files)]
for label, p in (("n = 1000", S.SketchParams.mash(1000, 1000, False, 21, 0)),
                 ("200 000 -> 1000 (CLI default)", S.SketchParams.mash(200_000, 1000, False, 21, 0))):
    H.sketch_files(distinct[:64], p, H.FilterParams(False))  # warm: handles, page cache
    for rep in range(2):
        t = time.time(); res = H.sketch_files(paths, p, H.FilterParams(False)); dt = time.time() - t
    print("%-30s %d x 5 Mb fasta: %.2f s, %.0f files/s, %.1f Gbases/s" % (label, n_files, dt, n_files / dt, n_files * 5e6 / dt / 1e9), flush=True)
