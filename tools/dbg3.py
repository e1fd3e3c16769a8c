import os, sys, time, threading
sys.path.insert(0, ".")
import numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
from finch_rs_amd.sketch_schemes import SketchParams
seq = S.synth_genome_host(200_000_000, 7).tobytes()
paths = []
for i in range(256):
    pth = "/tmp/e2e_%d.fa" % i
    with open(pth, "wb") as f:
        f.write(b">g\n"); s5 = seq[(i % 60) * 3_000_000:(i % 60 + 1) * 3_000_000 + 2_000_000]
        f.write(b"\n".join(s5[j:j + 70] for j in range(0, len(s5), 70))); f.write(b"\n")
    paths.append(pth)
datas = [open(p, "rb").read() for p in paths]
def scan_worker(idx, nt):
    for i in range(idx, 256, nt): H.fastx_scan(datas[i])
for nt in (1, 4, 8, 16, 32):
    th = [threading.Thread(target=scan_worker, args=(i, nt)) for i in range(nt)]
    t = time.time(); [x.start() for x in th]; [x.join() for x in th]; dt = time.time() - t
    print("host scan only, %d threads: %.2f s  %.0f files/s" % (nt, dt, 256 / dt))
p = SketchParams.mash(1000, 1000, True, 21, 0)
for nt in (1, 4, 8, 16, 32):
    t = time.time(); res = H.sketch_files(paths, p, H.FilterParams(False), n_threads=nt); dt = time.time() - t
    print("sketch_files, %d threads: %.2f s  %.0f files/s" % (nt, dt, 256 / dt))
