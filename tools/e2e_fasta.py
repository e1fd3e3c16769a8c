#!/usr/bin/env python3
"""One large multi-line FASTA (a 1.2 Gb synthetic genome in ~30 records, 70-column lines) in host memory -> sketch: the FASTA leg
of the device-side text path end to end (GPU box).  python tools/e2e_fasta.py [Mbases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import finch_rs_amd as F
from finch_rs_amd import host as H, sketch_schemes as S
from oracle import oracle as O
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
L = mb * 1_000_000
dg = F.DeviceBuffer(L)
S.synth_genome_device(dg, L, 77)
g = dg.download(L)
recs = []
per = L // 30
for i in range(30):
    seq = g[i * per:(i + 1) * per]
    rows = (len(seq) + 69) // 70
    a = np.full((rows, 71), 10, np.uint8)
    pad = np.zeros(rows * 70, np.uint8); pad[:len(seq)] = seq
    a[:, :70] = pad.reshape(rows, 70)
    last = len(seq) - (rows - 1) * 70
    recs.append(b">chr%d\n" % i + a.reshape(-1)[:(rows - 1) * 71 + last].tobytes() + b"\n")
data = np.frombuffer(b"".join(recs), np.uint8)
p = F.SketchParams.mash(1000, 1000, True, 21, 0)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); res = H.sketch_stream(data, "g", p, H.FilterParams(False)); best = min(best, time.perf_counter() - t0)
print("FASTA %.2f GB: %.1f ms  %.1f GB/s of text  %.1f Gbases/s" % (data.size / 1e9, best * 1e3, data.size / best / 1e9, 30 * per / best / 1e9))
if mb <= 300:  # the oracle takes ~10 s per 300 Mb
    o = O.OracleSketcher(O.MASH, 1000, 21, 0)
    assert o.sketch_stream(data.tobytes()) == 1
    assert np.array_equal(res.sketch(0).arrays[0], o.to_vec()[0]) and np.array_equal(res.sketch(0).arrays[1], o.to_vec()[1])
    print("equal to the oracle's sketch")
