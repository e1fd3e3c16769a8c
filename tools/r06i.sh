cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v "launch \|range \|finish:" | tee gpurun_out/r06i_e2e_trace.txt
import time, numpy as np, sys
sys.path.insert(0, ".")
import finch_rs_amd as F
from finch_rs_amd import host as H, sketch_schemes as S
ns, RL = 4_000_000, 150
g = S.synth_genome_host(5_000_000, 20250620)
reads = S.synth_reads_host(g, 0, ns, RL, 20250620, 10000, 500).reshape(ns, RL + 1)[:, :RL]
w = 12 + RL + 3 + RL + 1
txt = np.empty((ns, w), np.uint8)
txt[:, 0], txt[:, 1] = ord("@"), ord("r")
idx = np.arange(ns, dtype=np.int64)
for d in range(9):
    txt[:, 10 - d] = 48 + (idx // 10 ** d) % 10
txt[:, 11] = 10
txt[:, 12:12 + RL] = reads
txt[:, 12 + RL:15 + RL] = np.frombuffer(b"\n+\n", np.uint8)
txt[:, 15 + RL:15 + 2 * RL] = ord("I")
txt[:, w - 1] = 10
data = txt.reshape(-1)
p = F.SketchParams.mash(1000, 1000, True, 21, 0)
for thr, chunk in ((None, None), ("4", None), ("16", "33554432"), ("12", None), ("16", "268435456")):
    F.debug_set(trace="1", read_threads=thr, fastq_strip_chunk=chunk)
    for _ in range(3):
        t0 = time.perf_counter()
        res = H.sketch_stream(data, "fastq", p, H.FilterParams(False))
        dt = time.perf_counter() - t0
    print("read_threads=%s chunk=%s: %.1f ms %.2f Gbases/s" % (thr, chunk, dt * 1e3, ns * RL / dt / 1e9), flush=True)
PY
