"""Print the kernel timeline (start ms, duration ms, name) of a rocprofv3 --kernel-trace csv, kernels >= --min-ms or k2_sketch.
usage: python tools/kernel_timeline.py <dir-or-csv> [--min-ms 0.1] [--skip N]"""
import csv, glob, os, sys
src = sys.argv[1]
min_ms = float(sys.argv[sys.argv.index("--min-ms") + 1]) if "--min-ms" in sys.argv else 0.1
f = src if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    n = r["Kernel_Name"]
    if "synth" in n: continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    if (e - s) / 1e6 >= min_ms or "k2_sketch" in n: print("%9.3f %8.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n[:60]))
