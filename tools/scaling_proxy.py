#!/usr/bin/env python3
"""What ONE GPU predicts for the 1/2/4/8 curve of BASELINE configs[3] (50 Gbase in total, strong scaling): each rank's share
(50 / N Gbase) alone on the GPU, one handle, as `python bench.py --gbases <share>` runs it; predicted efficiency at N =
t(50) / (N t(50 / N)).  Then N handles on the one GPU through the driver-shaped launches.   (GPU box)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "20"


def run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode not in (0,) or not lines:
        print("bench failed:", r.stderr[-500:])
        return None
    return json.loads(lines[-1])


t = {}
for n in (1, 2, 4, 8):
    d = run(["--gbases", str(50.0 / n), "--steps", steps, "--warmup", "5", "--no-extras", "--no-cpu-baseline", "--no-live-pmc"])
    if d:
        t[n] = d["ms_per_step"]
        print("share 50/%d = %6.3f Gbase: %8.3f ms per pass  %6.1f Gbases/s  kernel %.3f ms  (%s)" %
              (n, 50.0 / n, d["ms_per_step"], d["value"] / 1e9, d["roofline"]["kernel_ms_per_pass"], d["roofline"]["kernel"][:24]), flush=True)
for n in (2, 4, 8):
    if 1 in t and n in t:
        print("predicted efficiency at N = %d: %.3f" % (n, t[1] / (n * t[n])))
for n in (2, 4, 8):
    d = run(["--gpus", str(n), "--share-gpu", "--steps", "5", "--warmup", "2", "--no-extras", "--no-cpu-baseline"])
    if d:
        print("N = %d handles on the one GPU (one fh_sketch_device_blocks call per step): %8.3f ms per step  golden %s  per-rank kernel ms %s" %
              (n, d["ms_per_step"], d["sketch_check"]["matches_golden"], [p["kernel_ms_per_pass"] for p in d["per_rank"]]), flush=True)
