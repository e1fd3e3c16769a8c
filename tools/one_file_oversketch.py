"""one 5 Mb FASTA, CLI-default oversketch, a few times on one worker thread (for rocprofv3 / FH_DEBUG=trace)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finch_rs_amd import host as H, sketch_schemes as S
pth = "/tmp/one.fa"
if not os.path.exists(pth):
    s5 = S.synth_genome_host(5_000_000, 7).tobytes()
    with open(pth, "wb") as f:
        f.write(b">g\n"); f.write(b"\n".join(s5[j:j + 70] for j in range(0, len(s5), 70))); f.write(b"\n")
p = S.SketchParams.mash(int(sys.argv[1]) if len(sys.argv) > 1 else 200_000, 1000, False, 21, 0)
for rep in range(6):
    t = time.time(); res = H.sketch_files([pth, pth], p, H.FilterParams(False), n_threads=1); dt = time.time() - t
    print("2 files on one worker: %.2f ms per file" % (dt * 500), flush=True)
