import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import finch_rs_amd as F
from finch_rs_amd import host as H, sketch_schemes as S
ns = 4_000_000; RL, REC = 150, 151
dg = F.DeviceBuffer(5_000_000); dr = F.DeviceBuffer(ns * REC + 64)
S.synth_genome_device(dg, 5_000_000, 20250620); S.synth_reads_device(dr, dg, 5_000_000, 0, ns, RL, 20250620, 10000, 500)
reads = dr.download(ns * REC).reshape(ns, REC)[:, :RL]
w = 12 + RL + 3 + RL + 1
txt = np.empty((ns, w), np.uint8); txt[:, 0], txt[:, 1] = ord("@"), ord("r")
idx = np.arange(ns, dtype=np.int64)
for d in range(9): txt[:, 10 - d] = 48 + (idx // 10 ** d) % 10
txt[:, 11] = 10; txt[:, 12:12 + RL] = reads; txt[:, 12 + RL:15 + RL] = np.frombuffer(b"\n+\n", np.uint8); txt[:, 15 + RL:15 + 2 * RL] = ord("I"); txt[:, w - 1] = 10
data = txt.reshape(-1)
p = F.SketchParams.mash(1000, 1000, True, 21, 0)
def run(label, fn):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t0)
    print("%-28s %.1f ms  %.1f GB/s of text" % (label, best * 1e3, data.size / best / 1e9), flush=True)
    return r
a = run("one handle", lambda: H.sketch_stream(data, "x", p, H.FilterParams(False)))
for nd in (1, 2, 4):
    b = run("sharded, %d handle(s) on dev 0" % nd, lambda: H.sketch_stream_sharded(data, "x", p, H.FilterParams(False), [0] * nd))
    assert np.array_equal(a.sketch(0).arrays[0], b.sketch(0).arrays[0])
