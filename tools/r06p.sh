cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for dbg in "trace" "trace,no_numa_pin"; do
echo "== FH_DEBUG=$dbg"
FH_DEBUG=$dbg python - <<'PY' 2>&1 | grep "fastq host strip\|Gbases" | cut -c1-220
import time, numpy as np, sys
sys.path.insert(0, ".")
import torch
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
best = 1e9
for _ in range(4):
    t0 = time.perf_counter()
    res = H.sketch_stream(data, "fastq", p, H.FilterParams(False))
    best = min(best, time.perf_counter() - t0)
print("best %.1f ms %.2f Gbases/s" % (best * 1e3, ns * RL / best / 1e9))
PY
done
done 2>&1 | tee gpurun_out/r06p_numa_pin.txt
