"""SURVEY config C5 by count on one GPU: 10 000 FASTA inputs of 5 Mb through finch_sketch_files in one call (the list
cycles over 256 distinct files so the box's page cache, not its disk, feeds it).  Reports files/s, checks that every
repeat of a file gives the same sketch and that host RSS / device memory do not grow with the file count.
usage (GPU box): python tools/batch_c5.py [n_files]"""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finch_rs_amd import host as H, sketch_schemes as S
n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
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
p = S.SketchParams.mash(1000, 1000, False, 21, 0)
H.sketch_files(distinct, p, H.FilterParams(False))  # warm: handles, page cache
def mem():
    free, total = torch.cuda.mem_get_info(0)
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, (total - free) / 2**20
rss0, dev0 = mem()
t = time.time(); res = H.sketch_files(paths, p, H.FilterParams(False)); dt = time.time() - t
rss1, dev1 = mem()
assert len(res) == n_files
ref = {}
bad = 0
for i, pth in enumerate(paths):
    if i >= 256 and i % 16: continue  # every file once, then a sample of the repeats (Python-side copy is the slow part)
    kc, km = res.sketch(i).arrays
    key = (kc.tobytes(), km.tobytes())
    if ref.setdefault(pth, key) != key: bad += 1
print("%d x 5 Mb fasta in one call: %.2f s, %.0f files/s, %.1f Gbases/s; repeats differing: %d; host RSS %.0f -> %.0f MiB, device used %.0f -> %.0f MiB"
      % (n_files, dt, n_files / dt, n_files * 5e6 / dt / 1e9, bad, rss0, rss1, dev0, dev1), flush=True)
assert bad == 0
