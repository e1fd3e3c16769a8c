import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F  # noqa: E402
F.debug_set(trace="1")
seq = S.synth_genome_host(20_000_000, 7).tobytes()
paths = []
for i in range(24):
    pth = "/dev/shm/ow_%d.fa" % i
    with open(pth, "wb") as f:
        s5 = seq[(i % 3) * 5_000_000:(i % 3 + 1) * 5_000_000]
        f.write(b">g\n"); f.write(b"\n".join(s5[j:j + 70] for j in range(0, len(s5), 70))); f.write(b"\n")
    paths.append(pth)
p = S.SketchParams.mash(1000, 1000, False, 21, 0)
H.sketch_files(paths[:4], p, H.FilterParams(False), n_threads=1)
print("---- timed ----", flush=True)
t = time.time(); H.sketch_files(paths[:6], p, H.FilterParams(False), n_threads=1); dt = time.time() - t
print("6 files, one worker: %.2f ms per file" % (dt / 6 * 1e3))
for pth in paths: os.remove(pth)
