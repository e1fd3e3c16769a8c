"""plain gzip through finch_sketch_files, device-side inflate against the host's:
    python tools/gz_bench.py [reads [level [noisy|const [substitutions per million]]]]      (on an MI355X)
1 M reads of 150 bases from a 5 Mb genome (30-fold coverage), 1 % substitutions unless told otherwise (0: every k-mer thirty
times over -- the case in which a speculative first threshold falls short)."""
import os, sys, time, zlib, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finch_rs_amd as F
from finch_rs_amd import host as H, sketch_schemes as S

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
noisy = len(sys.argv) > 3 and sys.argv[3] == "noisy"
RL = 150
g = S.synth_genome_host(5_000_000, 1)
sub_ppm = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000
reads = S.synth_reads_host(g, 0, ns, RL, 7, sub_ppm, 500).reshape(ns, RL + 1)[:, :RL]
w = 12 + RL + 3 + RL + 1
txt = np.empty((ns, w), np.uint8)
txt[:, 0], txt[:, 1] = ord("@"), ord("r")
idx = np.arange(ns)
for d in range(9):
    txt[:, 10 - d] = 48 + (idx // 10 ** d) % 10
txt[:, 11] = 10
txt[:, 12:12 + RL] = reads
txt[:, 12 + RL:15 + RL] = np.frombuffer(b"\n+\n", np.uint8)
txt[:, 15 + RL:15 + 2 * RL] = np.random.default_rng(1).integers(35, 74, size=(ns, RL), dtype=np.uint8) if noisy else ord("I")
txt[:, w - 1] = 10
raw = txt.tobytes()
d = tempfile.mkdtemp(prefix="gzb_", dir="/dev/shm")
try:
    path = os.path.join(d, "r.fastq.gz")
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    open(path, "wb").write(co.compress(raw) + co.flush())
    print("text %.1f MB, gzip level %d: %.1f MB (x%.2f)%s" % (len(raw) / 1e6, level, os.path.getsize(path) / 1e6, len(raw) / os.path.getsize(path), " noisy quals" if noisy else ""))
    p = F.SketchParams.mash(1000, 1000, True, 21, 0)
    res = {}
    modes = (("device", None), ("host", "0"))
    if os.environ.get("GZ_ONLY"):
        modes = tuple(m for m in modes if m[0] == os.environ["GZ_ONLY"])
    for name, env in modes:
        if env:
            F.debug_set(device_gzip=env)
        best = 1e9
        for _ in range(int(os.environ.get("GZ_REPS", "4"))):
            t0 = time.perf_counter()
            sk = H.sketch_files([path], p, H.FilterParams(False))
            best = min(best, time.perf_counter() - t0)
        F.debug_set(device_gzip=None)
        s0 = sk.sketch(0)
        res[name] = (s0.arrays[0].tobytes(), s0.seq_length)
        print("%-6s %.1f ms  %.2f Gbases/s  %.2f GB/s of text   on device / reread: %s" % (name, best * 1e3, ns * RL / best / 1e9, len(raw) / best / 1e9, H.debug_device_gzip()))
    assert len(res) < 2 or res["device"] == res["host"]
    if os.environ.get("GZ_TRACE"):
        F.debug_set(trace="1")
        H.sketch_files([path], p, H.FilterParams(False))
finally:
    shutil.rmtree(d, ignore_errors=True)
