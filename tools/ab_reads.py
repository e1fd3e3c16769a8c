#!/usr/bin/env python3
"""Segment kernels against the tile kernels on reads of a given length, one process, same resident stream:
fh_set_record_stride(1) = "do not use the segment kernels" (k2_sketch / k2_sketch_w) against the stride L + 1.
usage (GPU box): python tools/ab_reads.py [--len 250] [--gbases 4] [--ks 21,31] [--n 1000] [--steps 5] [--seed 0] [--ragged LO,HI]
--ragged LO,HI: records of every length LO..HI (uniform) instead of one length (host-generated; the stride hint is then 0 = ask
the block, against 1)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finch_rs_amd as F  # noqa: E402
from finch_rs_amd import sketch_schemes as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--len", type=int, default=250)
ap.add_argument("--gbases", type=float, default=4.0)
ap.add_argument("--ks", default="21,31")
ap.add_argument("--n", type=int, default=1000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--ragged", default="")
ap.add_argument("--frac-full", type=float, default=0.0, help="--ragged: this fraction of the records has the full length HI")
a = ap.parse_args()

GL = 5_000_000
gen = S.DeviceBuffer(GL + 64)
S.synth_genome_device(gen, GL, 20250620)
if a.ragged:
    lo, hi = (int(x) for x in a.ragged.split(","))
    rng = np.random.default_rng(1)
    nrec = int(a.gbases * 1e9 / ((lo + hi) / 2))
    lens = rng.integers(lo, hi + 1, size=nrec)
    lens[rng.random(nrec) < a.frac_full] = hi
    # reads of `hi` bases from the device generator, cut to their lengths on the host (bounded: a few hundred Mbases)
    full = S.DeviceBuffer(nrec * (hi + 1) + 64)
    S.synth_reads_device(full, gen, GL, 0, nrec, hi, 20250620, 10000, 500)
    h = full.download(nrec * (hi + 1)).reshape(nrec, hi + 1)
    keep = np.arange(hi + 1)[None, :] < lens[:, None]
    h[~keep] = 255
    h[np.arange(nrec), lens] = 0
    stream = h[h != 255]
    full.free()
    total = len(stream)
    buf = S.DeviceBuffer(total + 256)
    buf.upload(stream)
    bases = int(lens.sum())
    hints = (("tile kernels", 1), ("segment kernels (block asked)", 0))
    what = "records of %d..%d bases (uniform%s), %.2f Gbases" % (lo, hi, ", %.0f %% at %d" % (100 * a.frac_full, hi) if a.frac_full else "", bases / 1e9)
else:
    L = a.len
    nrec = int(a.gbases * 1e9 / L)
    total = nrec * (L + 1)
    buf = S.DeviceBuffer(total + 256)
    S.synth_reads_device(buf, gen, GL, 0, nrec, L, 20250620, 10000, 500)
    bases = nrec * L
    hints = (("tile kernels", 1), ("segment kernels", L + 1))
    what = "%d reads of %d bases, %.2f Gbases" % (nrec, L, bases / 1e9)
print("resident stream: %s; n = %d, seed %d, best of %d passes each" % (what, a.n, a.seed, a.steps), flush=True)
for k in (int(x) for x in a.ks.split(",")):
    res = {}
    for name, hint in hints:
        sk = F.SketchParams.mash(a.n, a.n, True, k, a.seed).create_sketcher()
        sk.set_record_stride(hint)
        sk.set_profiling(True)
        best, kbest, fp = 1e30, 1e30, None
        for it in range(a.steps + 2):
            sk.reset()
            t0 = time.perf_counter()
            sk.push_device(buf.ptr, total)
            n, tk = sk.finish()
            dt = time.perf_counter() - t0
            kms = sk.kernel_time()[0]
            if it >= 2:
                best, kbest = min(best, dt), min(kbest, kms)
        kc, km, _ = sk.to_arrays()
        fp = (int(np.bitwise_xor.reduce(kc["hash"])), int(kc["count"].astype(np.uint64).sum()), int(km.astype(np.uint64).sum()), tk)
        res[name] = (best, kbest, fp, sk.debug_segments())
        sk.close()
    (n0, r0), (n1, r1) = list(res.items())
    assert r0[2] == r1[2], "the two kernels' sketches differ"
    print("k = %2d: %-30s %7.2f ms per pass (kernel %7.2f) %6.1f Gbases/s | %-30s %7.2f ms (kernel %7.2f) %6.1f Gbases/s | %+5.1f %%  seg launches %d stride %d, same sketch"
          % (k, n0, r0[0] * 1e3, r0[1], bases / r0[0] / 1e9, n1, r1[0] * 1e3, r1[1], bases / r1[0] / 1e9, 100.0 * (r0[0] / r1[0] - 1.0),
             r1[3][0], r1[3][2]), flush=True)
