#!/usr/bin/env python3
"""Fuzz of the batch sketcher (fh_batch_*, fh_k2b.hip) against the oracle: random k, n, seed, batch shapes and file contents
(random genomes, N-rich and lowercase text, many short records, repeats, tiny and empty files).  Every file the batch path
takes must carry the oracle's sketch bit for bit; every file it does not take is sketched through a HipSketcher and held
against the oracle as well.  usage: python tools/fuzz_batch.py [files=1500] [seed=1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finch_rs_amd as F  # noqa: E402
from oracle import oracle as O  # noqa: E402

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_file(rng):
    kind = rng.integers(0, 10)
    if kind == 0:
        return np.zeros(0, np.uint8)
    if kind == 1:  # tiny
        L = int(rng.integers(1, 200))
    elif kind == 2:  # around the sizes where the threshold appears / tiles end
        L = int(rng.choice([2047, 2048, 2049, 3999, 4000, 4001, 4097, 8191, 12000, 16384]))
    else:
        L = int(rng.integers(200, 300_000))
    if kind == 3:  # a repeat
        unit = rng.choice(ACGT, size=int(rng.integers(1, 400)))
        return np.concatenate([np.tile(unit, L // len(unit) + 1)[:L], np.zeros(1, np.uint8)])
    nrec = int(rng.integers(1, 1 + max(1, min(50, L // 20))))
    cuts = np.sort(rng.integers(0, L + 1, size=nrec - 1)) if nrec > 1 else np.zeros(0, np.int64)
    seq = rng.choice(ACGT, size=L)
    m = rng.random(L)
    p_n = float(rng.choice([0.0, 0.0005, 0.01, 0.2]))
    seq[m < p_n] = ord("N")
    p_low = float(rng.choice([0.0, 0.02, 0.5]))
    low = m > 1 - p_low
    seq[low] = seq[low] | 0x20
    if rng.random() < 0.1:
        seq[rng.random(L) < 0.01] = ord("U")
    parts, prev = [], 0
    for c in list(cuts) + [L]:
        parts.append(seq[prev:c])
        parts.append(np.zeros(1, np.uint8))
        prev = c
    return np.concatenate(parts)


def main():
    want = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    done = taken = batches = 0
    t0 = time.time()
    while done < want:
        k = int(rng.integers(1, 33))
        n = int(rng.choice([1, 10, 100, 500, 1000, 1000, 1000, 2000, 2500, 3000]))
        seed = int(rng.choice([0, 0, 42, 2**63 + 5]))
        nf = int(rng.integers(1, 40))
        blocks = [make_file(rng) for _ in range(nf)]
        b = F.BatchSketcher(n, k, seed, max_files=int(rng.choice([1, 3, 8, 64])), stage_bytes=int(rng.choice([1 << 20, 4 << 20])))
        two_bit = bool(rng.integers(0, 2))  # the link carries bytes, or the two-bit form (fh_batch_submit_packed)
        res = b.sketch_many(blocks, slot=int(rng.integers(0, 2)), two_bit=two_bit)
        batches += 1
        for i, (r, blk) in enumerate(zip(res, blocks)):
            ora = O.OracleSketcher(O.MASH, n, k, seed)
            ora.process_packed(blk, 0)
            okc, okm = ora.to_vec()
            otk = ora.total_bases_and_kmers()[1]
            if r is None:
                sk = F.SketchParams.mash(n, n, True, k, seed).create_sketcher()
                sk.push_block(blk)
                kc, km, _ = sk.to_arrays()
                tk = sk.finish()[1]
                sk.close()
            else:
                kc, km, _, tk = r
                taken += 1
            if not (np.array_equal(kc, okc) and np.array_equal(km, okm) and tk == otk):
                np.save("/tmp/fuzz_batch_fail.npy", blk)
                print("MISMATCH k=%d n=%d seed=%d file %d of %d (%d bytes, %s): %d vs %d hashes, total_kmers %d vs %d"
                      % (k, n, seed, i, nf, len(blk), "taken" if r is not None else "own sketcher", len(kc), len(okc), tk, otk))
                return 1
        b.close()
        done += nf
    print("fuzz_batch: %d files in %d batches, %d taken many-per-launch, %d through a sketcher of their own: all equal to the oracle (%.0f s)"
          % (done, batches, taken, done - taken, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
