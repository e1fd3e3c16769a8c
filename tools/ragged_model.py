#!/usr/bin/env python3
"""What a segment kernel for records of MANY lengths could save: a model of the hashed window slots per record under three
schemes, for a length distribution, k and the round length RO of fh_k2s.hip (65 - k for k <= 24, 32 beyond).

  tile      k2_sketch: one window slot per stream position (len + 1 per record)
  record    a lane per record, 64 consecutive records per wave: a round runs to the longest record's windows (wave maximum)
  items     the valid windows of a tile cut into work items of <= RO consecutive windows (runs cut at the lanes' block edges:
            every lane itemises its own TB / 64 positions), full items first, partial ones sorted by size, 64 items per round;
            + SETUP window-equivalents per item (a round's set-up is ~85 VALU instructions, a window ~56-65)

usage: python tools/ragged_model.py [--k 21] [--lo 35] [--hi 150] [--tile-records 64,128,256] [--reads 200000] [--frac-full 0.0]
--frac-full F: a fraction F of the reads has length hi (most reads untrimmed), the rest is uniform lo..hi."""
import argparse

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=21)
ap.add_argument("--lo", type=int, default=35)
ap.add_argument("--hi", type=int, default=150)
ap.add_argument("--tile-records", default="64,128,256")
ap.add_argument("--reads", type=int, default=200_000)
ap.add_argument("--frac-full", type=float, default=0.0)
a = ap.parse_args()
K = a.k
import os
RO = int(os.environ.get("RO", 65 - K if K <= 24 or K == 26 else 32))
SETUP = 85.0 / (56.0 if K <= 24 else 65.0)
rng = np.random.default_rng(1)
lens = rng.integers(a.lo, a.hi + 1, size=a.reads)
lens[rng.random(a.reads) < a.frac_full] = a.hi
win = np.maximum(lens - K + 1, 0)
tile_slots = float((lens + 1).sum())
valid = float(win.sum())
print("k = %d, RO = %d, reads %d..%d (%.0f %% at %d): %.1f windows per record, %.1f positions (%.1f %% of them windows)"
      % (K, RO, a.lo, a.hi, 100 * a.frac_full, a.hi, valid / a.reads, tile_slots / a.reads, 100 * valid / tile_slots))
print("  tile kernel: %.1f slots per record" % (tile_slots / a.reads))
# a lane per record
w64 = win[: a.reads // 64 * 64].reshape(-1, 64)
rec_slots = 0.0
for row_max in w64.max(axis=1):
    rounds = int(np.ceil(row_max / RO))
    rec_slots += 64 * row_max + 64 * SETUP * rounds
print("  a lane per record: %.1f slots per record (%+.1f %% against the tile kernel)" % (rec_slots / w64.size, 100 * (tile_slots / a.reads / (rec_slots / w64.size) - 1)))
# work items
for tr in (int(x) for x in a.tile_records.split(",")):
    pos = 0
    # positions of a tile: records laid end to end with a breaker; valid-window bitmap
    n_tiles = 0
    slots = 0.0
    slots_c = 0.0
    i = 0
    while i + tr <= a.reads and n_tiles < 400:
        L = lens[i:i + tr]
        tb = int((L + 1).sum())
        W = np.zeros(tb, bool)
        p = 0
        for l in L:
            if l >= K:
                W[p:p + l - K + 1] = True
            p += l + 1
        block = -(-tb // 64)
        items = []
        for b in range(0, tb, block):
            seg = W[b:b + block]
            # runs of ones in seg
            d = np.diff(np.concatenate(([0], seg.view(np.int8), [0])))
            for s, e in zip(np.flatnonzero(d == 1), np.flatnonzero(d == -1)):
                n = e - s
                items += [RO] * (n // RO)
                if n % RO:
                    items.append(n % RO)
        items = np.sort(np.array(items))[::-1]
        for r in range(0, len(items), 64):
            slots += 64 * (items[r] + SETUP)
        # "trimmed cells": every lane's block cut into cells of RO positions on a fixed grid, a cell = one item from its first
        # to its last valid window (holes inside stay), empty cells dropped, items sorted by size
        cells = []
        for b in range(0, tb, block):
            for c in range(b, min(b + block, tb), RO):
                seg = W[c:min(c + RO, b + block, tb)]
                nz = np.flatnonzero(seg)
                if len(nz):
                    cells.append(nz[-1] - nz[0] + 1)
        cells = np.sort(np.array(cells))[::-1]
        for r in range(0, len(cells), 64):
            slots_c += 64 * (cells[r] + SETUP)
        i += tr
        n_tiles += 1
    per = slots / (n_tiles * tr)
    print("  work items, tiles of %3d records (%.1f KB of text per wave): %.1f slots per record (%+.1f %% against the tile kernel)"
          % (tr, tb / 1024.0, per, 100 * (tile_slots / a.reads / per - 1)))
    per_c = slots_c / (n_tiles * tr)
    print("     ... as trimmed cells of a fixed grid (no run logic):                %.1f slots per record (%+.1f %%)"
          % (per_c, 100 * (tile_slots / a.reads / per_c - 1)))
