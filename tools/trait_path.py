#!/usr/bin/env python3
"""The drop-in path at the trait level (bench.py extras.trait_path): one fh_process call per 150-base record of
host memory, single thread, to_vec included -- and the same bytes as 32 MB blocks.   python tools/trait_path.py [reads]   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
RL, REC, GL, SEED = 150, 151, 5_000_000, 20250620
dg = F.DeviceBuffer(GL); dr = F.DeviceBuffer(ns * REC + 64)
S.synth_genome_device(dg, GL, SEED); S.synth_reads_device(dr, dg, GL, 0, ns, RL, SEED, 10000, 500)
reads = np.ascontiguousarray(dr.download(ns * REC))
offs = np.arange(ns, dtype=np.uint64) * REC
lens = np.full(ns, RL, dtype=np.uint64)
s = F.SketchParams.mash(1000, 1000, True, 21, 0).create_sketcher()
for name in ("records", "blocks"):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); s.reset()
        if name == "records":
            s.process_records(reads, offs, lens); t_calls = time.perf_counter() - t0
        else:
            blk = (32 << 20) // REC * REC
            for o in range(0, reads.size, blk):
                s.push_block(reads[o:o + blk])
        kc, km, _ = s.to_arrays(); tk = s.finish()[1]
        best = min(best, time.perf_counter() - t0)
    print("%-8s %.4f s  %.2f GB/s of sequence  (%.2f ns per record)  xor %x  segments %s" %
          (name, best, ns * RL / best / 1e9, best / ns * 1e9, int(np.bitwise_xor.reduce(kc["hash"])), s.debug_segments()), flush=True)
    if name == "records":
        print("         (last repeat: the calls alone %.4f s, reset + to_vec + finish behind them %.4f s)" %
              (t_calls, time.perf_counter() - t0 - t_calls), flush=True)
