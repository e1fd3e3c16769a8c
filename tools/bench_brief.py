#!/usr/bin/env python3
"""A bench.py JSON line in a few lines of text (for the tail gpurun prints): python tools/bench_brief.py <file.json>"""
import json
import sys

d = None
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
if d is None:
    sys.exit("no JSON line in %s" % sys.argv[1])
r = d.get("roofline") or {}
c = d.get("cpu_baseline") or {}
print("%s: %.1f Gbases/s  %.3f ms/step  N=%d  golden %s  roofline %.1f GB/s frac %s kernel %.3f ms/pass  cpu %s (%s cores)" % (
    (d.get("config") or {}).get("workload", "?")[:40], d["value"] / 1e9, d["ms_per_step"], d["n_gpus"],
    (d.get("sketch_check") or {}).get("matches_golden"), r.get("achieved") or 0, r.get("frac"), r.get("kernel_ms_per_pass") or 0,
    c.get("value"), c.get("cores")))
for k, v in (d.get("extras") or {}).items():
    print("  ", k, {kk: vv for kk, vv in v.items() if kk not in ("what", "pmc", "sketch_check", "roofline", "cpu_baseline")},
          (v.get("sketch_check") or {}).get("matches_golden"))
