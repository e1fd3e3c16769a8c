"""many gzip'd FASTQ files through ONE finch_sketch_files call, device-side inflate against the host's:
    python tools/gz_batch.py [files [reads per file]]      (on an MI355X)
Every worker thread of the call has a sketcher of its own, and with the device-side inflate each of them keeps a launch
waiting for its file's bytes: this is the check that they do not stand in each other's way."""
import os, sys, time, zlib, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finch_rs_amd as F
from finch_rs_amd import host as H, sketch_schemes as S

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
RL = 150
d = tempfile.mkdtemp(prefix="gzbatch_", dir="/dev/shm")
try:
    paths, total = [], 0
    for i in range(nf):
        g = S.synth_genome_host(2_000_000, 100 + i)
        reads = S.synth_reads_host(g, 0, ns, RL, i, 10_000, 500).reshape(ns, RL + 1)[:, :RL]
        w = 12 + RL + 3 + RL + 1
        txt = np.empty((ns, w), np.uint8)
        txt[:, 0], txt[:, 1] = ord("@"), ord("r")
        idx = np.arange(ns)
        for k in range(9):
            txt[:, 10 - k] = 48 + (idx // 10 ** k) % 10
        txt[:, 11] = 10
        txt[:, 12:12 + RL] = reads
        txt[:, 12 + RL:15 + RL] = np.frombuffer(b"\n+\n", np.uint8)
        txt[:, 15 + RL:15 + 2 * RL] = np.random.default_rng(i).integers(35, 74, size=(ns, RL), dtype=np.uint8)
        txt[:, w - 1] = 10
        p = os.path.join(d, "f%04d.fastq.gz" % i)
        co = zlib.compressobj(1, zlib.DEFLATED, 31)
        blob = co.compress(txt.tobytes()) + co.flush()
        open(p, "wb").write(blob)
        paths.append(p)
        total += len(blob)
    print("%d files of %d reads, %.1f MB of gzip in all (%.1f MB each)" % (nf, ns, total / 1e6, total / 1e6 / nf))
    prm = F.SketchParams.mash(1000, 1000, True, 21, 0)
    res = {}
    for name, env in (("device", None), ("host", "0")):
        if env:
            F.debug_set(device_gzip=env)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            r = H.sketch_files(paths, prm, H.FilterParams(False))
            best = min(best, time.perf_counter() - t0)
        F.debug_set(device_gzip=None)
        res[name] = [r.sketch(i).arrays[0].tobytes() for i in range(nf)]
        print("%-6s %.1f ms  %.0f files/s  %.2f Gbases/s   on device / reread: %s" % (name, best * 1e3, nf / best, nf * ns * RL / best / 1e9, H.debug_device_gzip()))
    assert res["device"] == res["host"]
finally:
    shutil.rmtree(d, ignore_errors=True)
