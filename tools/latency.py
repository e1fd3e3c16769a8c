"""Cold-call latency of finch_sketch_files on one small genome (BASELINE configs[0] shape) and of creating / freeing a
sketcher.  usage (GPU box): python tools/latency.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F
seq = S.synth_genome_host(5_000_000, 7).tobytes()
pth = "/tmp/c1.fa"
with open(pth, "wb") as f:
    f.write(b">g\n"); f.write(b"\n".join(seq[j:j + 70] for j in range(0, len(seq), 70))); f.write(b"\n")
p = S.SketchParams.mash(1000, 1000, False, 21, 0)
for rep in range(4):
    t = time.perf_counter(); res = H.sketch_files([pth], p, H.FilterParams(False)); dt = time.perf_counter() - t
    print("sketch_files(one 5 Mb FASTA): %.2f ms" % (dt * 1e3))
for rep in range(3):
    t = time.perf_counter(); sk = p.create_sketcher(); t1 = time.perf_counter(); sk.close(); t2 = time.perf_counter()
    print("fh_new (default in-flight) %.2f ms, fh_free %.2f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3))
for rep in range(3):
    t = time.perf_counter(); sk = p.create_sketcher(max_launch=2 << 20, stage_bytes=16 << 20); t1 = time.perf_counter(); sk.close(); t2 = time.perf_counter()
    print("fh_new (2 M positions in flight) %.2f ms, fh_free %.2f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3))
blk = np.frombuffer(seq + b"\x00", dtype=np.uint8)
for rep in range(3):
    t = time.perf_counter(); sk = p.create_sketcher(); sk.push_block(blk); n, tk = sk.finish(); t1 = time.perf_counter(); sk.close()
    print("fh_new + fh_push_block(5 MB) + fh_finish (the Rust binding's sequence for one small file): %.2f ms" % ((t1 - t) * 1e3))
