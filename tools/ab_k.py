#!/usr/bin/env python3
"""A/B of libfinch_hip builds on the resident synthetic stream (GPU box).

    python tools/ab_k.py --libs name=path.so[@K=V+K=V][,name=path.so...] --ks 21,31 [--n 1000] [--gbases 10] [--steps 3] [--env "K=V ..."]

Every (build, k) runs in its own process (FH_LIB picks the library): `steps` passes over the same `gbases` Gbase of SURVEY 8d M4
reads resident in HBM, best pass reported, with the kernel's own HIP-event time.  The fingerprints of the sketches are compared
across builds and -- for the sizes tests/golden/config_fingerprints.json holds -- with the oracle's.  Builds are made with e.g.
    FH_OUT=/root/repo/gpurun_out/lib_r8.so FH_EXTRA_FLAGS="-DFH_ROUND_BIG=8" python finch_rs_amd/csrc/build.py --force
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED, GL, RL = 20250620, 5_000_000, 150


def child(args):
    sys.path.insert(0, ROOT)
    import numpy as np
    import finch_rs_amd as F
    from finch_rs_amd import sketch_schemes as S
    n_reads = int(np.ceil(args.gbases * 1e9 / RL))
    rec = RL + 1
    dg = F.DeviceBuffer(GL)
    dr = F.DeviceBuffer(n_reads * rec + 64)
    S.synth_genome_device(dg, GL, SEED)
    S.synth_reads_device(dr, dg, GL, 0, n_reads, RL, SEED, 10000, 500)
    for k in [int(x) for x in args.ks.split(",")]:
        p = F.SketchParams.mash(args.n, args.n, True, k, 0)
        s = p.create_sketcher()
        s.set_profiling(True)
        best, kms, kp, kl = 1e30, 0.0, 0, 0
        for it in range(args.steps + 1):
            t0 = time.perf_counter()
            s.reset()
            s.push_device(dr.ptr, n_reads * rec)
            kc, km, _ = s.to_arrays()
            tk = s.finish()[1]
            dt = time.perf_counter() - t0
            ms, nl, npos = s.kernel_time()
            if it:
                best = min(best, dt)
                kms += ms; kp += npos; kl += nl
        fp = {"n_hashes": int(len(kc)), "hash_xor": int(np.bitwise_xor.reduce(kc["hash"])) if len(kc) else 0,
              "count_sum": int(kc["count"].astype(np.uint64).sum()), "extra_sum": int(kc["extra_count"].astype(np.uint64).sum()),
              "kmer_byte_sum": int(km.astype(np.uint64).sum()), "total_kmers": int(tk)}
        print("AB " + json.dumps({"k": k, "ms": round(best * 1e3, 3), "gbases_per_s": round(n_reads * RL / best / 1e9, 1),
                                  "kernel_GBps": round(kp / 1e9 / (kms / 1e3), 1) if kms else None, "launches": kl / args.steps, "fp": fp}), flush=True)
        s.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default=None)
    ap.add_argument("--ks", default="21,31")
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--gbases", type=float, default=10.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--env", default="")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    try:
        golden = json.load(open(os.path.join(ROOT, "tests", "golden", "config_fingerprints.json")))
    except Exception:
        golden = {}
    libs = [x.split("=", 1) for x in args.libs.split(",")] if args.libs else [["default", ""]]
    res = {}
    for name, path in libs:
        env = dict(os.environ)
        if path:
            env["FH_LIB"] = os.path.abspath(path)
            env["FH_NO_AUTOBUILD"] = "1"
        for kv in args.env.split():
            k, v = kv.split("=", 1)
            env[k] = v
        if "@" in path:  # name=path@K=V+K2=V2 : environment of this build alone (A/B of a run-time knob of one library)
            path, extra = path.split("@", 1)
            for kv in extra.split("+"):
                k, v = kv.split("=", 1)
                env[k] = v
            env.pop("FH_LIB", None)
            if path:
                env["FH_LIB"] = os.path.abspath(path)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--ks", args.ks, "--n", str(args.n), "--gbases", str(args.gbases),
                              "--steps", str(args.steps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        rows = [json.loads(l[3:]) for l in out.stdout.splitlines() if l.startswith("AB ")]
        if out.returncode != 0 or not rows:
            print("%s: FAILED rc=%d\n%s" % (name, out.returncode, out.stdout[-2000:]))
        res[name] = {r["k"]: r for r in rows}
    ks = [int(x) for x in args.ks.split(",")]
    print("%-18s" % "build" + "".join("  k=%-2d Gb/s (kern GB/s)" % k for k in ks))
    ok = True
    for name, _ in libs:
        line = "%-18s" % name
        for k in ks:
            r = res[name].get(k)
            line += "  %9.1f (%7.1f)    " % (r["gbases_per_s"], r["kernel_GBps"] or 0) if r else "  %-24s" % "-"
        print(line + "  launches/pass " + " ".join("%.2f" % res[name][k]["launches"] for k in ks if k in res[name]))
    for k in ks:
        fps = {name: res[name][k]["fp"] for name, _ in libs if k in res[name]}
        ref = None
        g = golden.get("c2_k%d_n%d" % (k, args.n)) if args.gbases == 10.0 else None
        if g:
            ref = {x: g[x] for x in ("n_hashes", "hash_xor", "count_sum", "extra_sum", "kmer_byte_sum", "total_kmers")}
        for name, fp in fps.items():
            if ref is None:
                ref = fp
            if fp != ref:
                ok = False
                print("MISMATCH k=%d %s: %s != %s" % (k, name, fp, ref))
        print("k=%d fingerprints: %s%s" % (k, "all equal" if all(fp == ref for fp in fps.values()) else "DIFFER", " (== oracle golden)" if g else ""))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
