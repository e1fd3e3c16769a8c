#!/usr/bin/env python3
"""A/B of library builds on the oversketch workloads (GPU box): configs[1]'s stream at (k, n) = (21, 200000), (31, 2000000), (21, 1000),
best of --reps passes each in its own process.   python tools/ab_oversketch.py name=path.so[,name=path.so...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(21, 1000), (21, 200000), (31, 2000000)]

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import time
    import numpy as np
    import finch_rs_amd as F
    from finch_rs_amd import sketch_schemes as S
    RL, REC, GL, SEED = 150, 151, 5_000_000, 20250620
    n_reads = int(np.ceil(10e9 / RL))
    dg = F.DeviceBuffer(GL); dr = F.DeviceBuffer(n_reads * REC + 64)
    S.synth_genome_device(dg, GL, SEED); S.synth_reads_device(dr, dg, GL, 0, n_reads, RL, SEED, 10000, 500)
    for k, n in CASES:
        s = F.SketchParams.mash(n, n, True, k, 0).create_sketcher()
        best = 1e9
        for it in range(4):
            t0 = time.perf_counter(); s.reset(); s.push_device(dr.ptr, n_reads * REC); nn, tk = s.finish(); dt = time.perf_counter() - t0
            if it: best = min(best, dt)
        kc, km, _ = s.to_arrays()
        print("AB " + json.dumps({"k": k, "n": n, "ms": round(best * 1e3, 2), "fp": [int(np.bitwise_xor.reduce(kc["hash"])), int(kc["count"].astype(np.uint64).sum()),
                                                                             int(kc["extra_count"].astype(np.uint64).sum()), int(km.astype(np.uint64).sum()), int(tk)]}), flush=True)
        s.close()
    sys.exit(0)

libs = [x.split("=", 1) for x in sys.argv[1].split(",")]
res = {}
for name, path in libs:
    env = dict(os.environ, FH_LIB=os.path.abspath(path), FH_NO_AUTOBUILD="1")
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "x"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    rows = [json.loads(l[3:]) for l in out.stdout.splitlines() if l.startswith("AB ")]
    if not rows:
        print(name, "FAILED", out.stdout[-1500:])
    res[name] = {(r["k"], r["n"]): r for r in rows}
print("%-10s" % "build" + "".join("  k=%d n=%-8d" % c for c in CASES))
for name, _ in libs:
    print("%-10s" % name + "".join("  %10.2f ms    " % res[name][c]["ms"] if c in res[name] else "   -" for c in CASES))
ok = all(len({json.dumps(res[name][c]["fp"]) for name, _ in libs if c in res[name]}) == 1 for c in CASES)
print("fingerprints:", "all builds equal" if ok else "DIFFER")
sys.exit(0 if ok else 1)
