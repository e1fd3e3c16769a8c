"""Throughput of finch_sketch_files over 256 x 5 Mb FASTA genomes by number of worker threads (SURVEY C5 shape, one
GPU).  usage (GPU box): [GPU_MAX_HW_QUEUES=8] python tools/batch_threads.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finch_rs_amd import host as H, sketch_schemes as S
seq = S.synth_genome_host(200_000_000, 7).tobytes()
paths = []
for i in range(256):
    pth = "/tmp/e2e_%d.fa" % i
    if not os.path.exists(pth):
        with open(pth, "wb") as f:
            f.write(b">g\n"); s5 = seq[(i % 60) * 3_000_000:(i % 60 + 1) * 3_000_000 + 2_000_000]
            f.write(b"\n".join(s5[j:j + 70] for j in range(0, len(s5), 70))); f.write(b"\n")
    paths.append(pth)
p = S.SketchParams.mash(1000, 1000, False, 21, 0)
for nt in (1, 2, 4, 6, 8, 12, 16):
    H.sketch_files(paths[:2 * nt], p, H.FilterParams(False), n_threads=nt)  # warm: handles, page cache
    t = time.time(); res = H.sketch_files(paths, p, H.FilterParams(False), n_threads=nt); dt = time.time() - t
    print("hw queues %s: 256 x 5 Mb fasta, %2d threads: %.3f s  %.0f files/s" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), nt, dt, 256 / dt), flush=True)
