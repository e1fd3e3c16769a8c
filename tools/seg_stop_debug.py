#!/usr/bin/env python3
"""Waves that stop inside tiles (loose threshold, few waves): total_kmers and hashes against the oracle.  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import finch_rs_amd as F
from oracle import oracle as O
rng = np.random.default_rng(77)
genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=400000)
recs = []
for _ in range(120):
    L = int(rng.integers(0, 5001)); st = int(rng.integers(0, len(genome) - L)); recs.append(bytes(genome[st:st + L]))
packed = np.frombuffer(b"".join(r + b"\0" for r in recs), dtype=np.uint8)
buf = F.DeviceBuffer(packed.size + 64); buf.upload(packed)
for k in (21, 31, 48):
    for inflight in (4096, 16384, 0):
        for stride in (1, 151, 100):
            p = F.SketchParams.scaled(1000, k, 0.5, 42 if k > 32 else 0)
            sk = p.create_sketcher(max_launch=inflight)
            sk.set_record_stride(stride)
            sk.push_device(buf.ptr, packed.size); sk.sync()
            kc, km, _ = sk.to_arrays(); tk = sk.finish()[1]
            ora = O.OracleSketcher(O.SCALED, 1000, k, 42 if k > 32 else 0, 0.5); ora.process_packed(packed, 0)
            okc, _ = ora.to_vec()
            print("k %2d inflight %6d stride %3d: hashes %6d / %6d  total_kmers %7d / %7d  %s  %s" % (
                k, inflight, stride, len(kc), len(okc), tk, ora.total_bases_and_kmers()[1],
                "OK" if len(kc) == len(okc) and tk == ora.total_bases_and_kmers()[1] else "DIFFERENT", sk.debug_counters()), flush=True)
            sk.close()
