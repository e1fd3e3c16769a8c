#!/usr/bin/env python3
"""Register / scratch use of the sketch kernels, from the code objects hipcc makes for gfx950 (no GPU needed).

    python tools/k2_regs.py                 # every K = 1..64 as the library is built (table -> stdout)
    python tools/k2_regs.py --k 31 [--asm out.s] [--flags "..."] [-D NAME=V ...]   # one K, development build

Prints per kernel: VGPRs, spilled VGPRs, scratch bytes per lane, SGPRs, LDS bytes (llvm-readelf --notes of the unbundled
device object).  profiles/r03_k2_registers.txt is this script's output.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "finch_rs_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-Wno-cuda-compat",
        "--cuda-device-only"]


def notes(elf):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], text=True)
    out = []
    for blk in txt.split("- .agpr_count:")[1:]:
        def g(key):
            m = re.search(r"\.%s:\s+(\S+)" % key, blk)
            return m.group(1) if m else "?"
        out.append({"name": g("name"), "vgpr": g("vgpr_count"), "spill": g("vgpr_spill_count"), "scratch": g("private_segment_fixed_size"),
                    "sgpr": g("sgpr_count"), "lds": g("group_segment_fixed_size")})
    return out


def compile_one(src, defs, flags, asm=None):
    d = tempfile.mkdtemp(prefix="k2regs_")
    co, elf = os.path.join(d, "k.co"), os.path.join(d, "k.elf")
    cmd = BASE + flags + ["-D%s" % x for x in defs] + ["-c", src, "-o", co]
    subprocess.check_call(cmd, cwd=CSRC)
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + co, "--output=" + elf])
    if asm:
        with open(asm, "w") as f:
            subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", elf], stdout=f)
    return notes(elf)


def pretty(name):
    m = re.match(r"_ZN2fh9k2_sketchILi(\d+)ELb(\d)ELb(\d)ELb(\d)EEE", name)
    if m:
        return "k2_sketch<%s,%s%s%s>" % (m.group(1), "M" if m.group(2) == "1" else "-", "S0" if m.group(3) == "1" else "--",
                                          "L" if m.group(4) == "1" else "-")
    m = re.match(r"_ZN2fh11k2_sketch_wILi(\d+)EEE", name)
    if m:
        return "k2_sketch_w<%s>" % m.group(1)
    return name


def main():
    sys.path.insert(0, CSRC)
    import importlib.util
    spec = importlib.util.spec_from_file_location("fh_build", os.path.join(CSRC, "build.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--asm", default=None)
    ap.add_argument("--flags", default=None, help="replaces the build's extra flags for the sketch kernel")
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--all-variants", action="store_true", help="list the masked / seeded / re-read variants too")
    args = ap.parse_args()
    kflags = args.flags.split() if args.flags is not None else list(B.K2_FLAGS)
    rows = []
    if args.k is not None:
        if args.k <= 32:
            rows = compile_one("fh_k2.hip", ["FH_PART=0", "FH_ONLY_K=%d" % args.k] + args.D, kflags, args.asm)
        else:
            rows = compile_one("fh_k2w.hip", ["FH_PART=%d" % ((args.k - 33) // 8)] + args.D, [], args.asm)
            rows = [r for r in rows if "ILi%dE" % args.k in r["name"]]
    else:
        jobs = [("fh_k2.hip", ["FH_PART=%d" % p], kflags) for p in range(B.NPARTS)]
        jobs += [("fh_k2w.hip", ["FH_PART=%d" % p], []) for p in range(B.NPARTS)]
        with ThreadPoolExecutor(max_workers=4) as ex:
            for r in ex.map(lambda j: compile_one(*j), jobs):
                rows += r
    rows = [r for r in rows if "k2_sketch" in r["name"]]
    if not args.all_variants:
        rows = [r for r in rows if "k2_sketch_w" in r["name"] or "ELb0ELb1ELb0E" in r["name"]]

    def key(r):
        m = re.search(r"ILi(\d+)E", r["name"])
        return (int(m.group(1)) if m else 0, r["name"])
    print("%-28s %5s %6s %8s %5s %6s" % ("kernel", "VGPR", "spill", "scratch", "SGPR", "LDS"))
    for r in sorted(rows, key=key):
        print("%-28s %5s %6s %8s %5s %6s" % (pretty(r["name"]), r["vgpr"], r["spill"], r["scratch"], r["sgpr"], r["lds"]))


if __name__ == "__main__":
    main()
