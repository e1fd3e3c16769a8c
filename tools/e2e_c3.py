"""configs[2] end to end through the host layer: FASTQ file -> k = 31 sketch of 2 M hashes (final size 10 000 x default
oversketch 200) -> strand / error / abundance filtering -> truncate -> Mash-JSON.  Where does the time go?
usage (GPU box): python tools/e2e_c3.py [n_reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
n_reads, rl = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000, 150
path = "/tmp/e2e.fastq"
if not os.path.exists(path):
    g = S.synth_genome_host(5_000_000, 1)
    reads = S.synth_reads_host(g, 0, n_reads, rl, 1, 10000, 500).reshape(n_reads, rl + 1)[:, :rl]
    with open(path, "wb") as f:
        for i in range(n_reads):
            f.write(b"@r%d\n" % i); f.write(reads[i].tobytes()); f.write(b"\n+\n"); f.write(b"I" * rl); f.write(b"\n")
size = os.path.getsize(path)
for label, p in (("n=1000", S.SketchParams.mash(1000, 1000, True, 31, 0)),
                 ("oversketch 2M -> 10000", S.SketchParams.mash(2_000_000, 10_000, True, 31, 0))):
    for filt in (H.FilterParams(False), H.FilterParams(None, (None, None), 0.0, 0.1)):  # off / FASTQ default + strand filter
        best = 1e9
        for rep in range(3):
            t = time.time(); res = H.sketch_files([path], p, filt); best = min(best, time.time() - t)
        t = time.time(); js = res.to_json(); tj = time.time() - t
        print("%-24s filter_on=%-5s: %.3f s (%.2f Gbases/s), %d hashes kept, to_json %.3f s (%d KB)"
              % (label, filt.filter_on, best, n_reads * rl / best / 1e9, res.sketch(0).arrays[0].shape[0], tj, len(js) // 1024), flush=True)
