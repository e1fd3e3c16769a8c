#!/usr/bin/env python3
"""What the GPU did during a rocprofv3 --kernel-trace run: span, union-busy time and the per-kernel totals of the last
`--tail` fraction of the trace (the timed call of tools/batch_trace.py), plus -- with --chain N -- the first N kernels in
start order with their start time, duration and the idle gap before each (the dependency chain of one file / one pass).
usage: python tools/trace_busy.py <dir-or-csv> [--tail 0.6] [--chain 40] [--skip-synth]"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
tail = float(sys.argv[sys.argv.index("--tail") + 1]) if "--tail" in sys.argv else 0.6
chain = int(sys.argv[sys.argv.index("--chain") + 1]) if "--chain" in sys.argv else 0
f = src if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)) if "synth" not in r["Kernel_Name"])
ev = ev[int(len(ev) * (1.0 - tail)):]
t0, t1 = ev[0][0], max(e for _, e, _ in ev)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in ev)
print("kernels %d  span %.2f ms  union-busy %.2f ms (%.0f %%)  sum of durations %.2f ms (overlap factor %.2f)"
      % (len(ev), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), tot / 1e6, tot / max(busy, 1)))
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    k = n.split("(")[0].replace("void ", "")
    per[k][0] += 1
    per[k][1] += e - s
for k, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %6d x %8.2f us = %9.3f ms  %s" % (c, d / c / 1e3, d / 1e6, k[:70]))
if chain:
    print("first %d kernels: start ms, duration us, gap before us, name" % chain)
    prev = None
    for s, e, n in ev[:chain]:
        print("  %9.3f %9.2f %9.2f  %s" % ((s - t0) / 1e6, (e - s) / 1e3, (s - prev) / 1e3 if prev is not None else 0.0, n.split("(")[0].replace("void ", "")[:60]))
        prev = e
