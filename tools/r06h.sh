cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_host_layer.py -x -q -k "fastq_text_in_memory or device_side_fastq" 2>&1 | tail -8
python - <<'PY' 2>&1 | tee gpurun_out/r06h_e2e_fastq.txt
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
for mode, thr in (("0", None), ("1", None), (None, None), (None, "8"), (None, "32"), ("1", "4")):
    F.debug_set(fastq_host_strip=mode, read_threads=thr)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        res = H.sketch_stream(data, "fastq", p, H.FilterParams(False))
        best = min(best, time.perf_counter() - t0)
    sk = res.sketch(0)
    print("fastq_host_strip=%s read_threads=%s: %.1f ms  %.2f Gbases/s  %.1f GB/s of text  (xor %x, host-stripped inputs so far %d)"
          % (mode, thr, best * 1e3, ns * RL / best / 1e9, data.size / best / 1e9, int(np.bitwise_xor.reduce(sk.arrays[0]["hash"])), H.debug_fastq_host_strip()), flush=True)
PY
