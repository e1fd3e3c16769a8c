#!/usr/bin/env python3
"""Register / scratch use of the sketch kernels, from the code objects hipcc makes for gfx950 (no GPU needed).

    python tools/k2_regs.py                 # every K = 1..64 as the library is built (table -> stdout)
    python tools/k2_regs.py --objects       # the same from the objects the shipped library was linked from (csrc/obj/*.o)
    python tools/k2_regs.py --k 31 [--asm out.s] [--flags "..."] [-D NAME=V ...]   # one K, development build

Prints per kernel: VGPRs, spilled VGPRs, scratch bytes per lane, SGPRs, LDS bytes (llvm-readelf --notes of the unbundled
device object).  profiles/r04_k2_registers.txt is this script's output on the shipped objects.  Exit code 1 if any sketch
kernel spills a register or needs more than 128 VGPRs (four waves per SIMD: fh_k2.hip).
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


def unbundle_object(obj):
    """device notes of an object file of the library build (a host object with the code object bundled inside)"""
    d = tempfile.mkdtemp(prefix="k2regs_")
    fat, elf = os.path.join(d, "fat.bin"), os.path.join(d, "k.elf")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + elf])
    return notes(elf)


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


FAST = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_mov_b32", "v_not_b32",
        "v_bitop3_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}  # ~2.4 cycles per wave64 instruction (profiles/r01_ubench_valu_rates.txt)


def hot_mix(asm_path, kernel_substr):
    """VALU / LDS instructions per position in the hot loop of a sketch kernel: the stretch of code between the first and the
    last per-position reject test (v_cmp_ge_u32 vcc, s.., v.. ; s_cbranch_vccnz -> the out-of-line admit path), divided by the
    number of tests in it.  Classes by issue cost on gfx950: fast ~2.4 cycles (FAST above), slow ~4.15 (everything else)."""
    txt = open(asm_path).read()
    for f in re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", txt):
        m = re.match(r"[0-9a-f]+ <([^>]+)>:", f)
        if not m or kernel_substr not in m.group(1):
            continue
        ops = [ln.split()[0] for ln in (x.strip() for x in f.split("\n")[1:]) if ln and not ln.startswith("//")]
        lines = [ln.strip() for ln in f.split("\n")[1:] if ln.strip() and not ln.strip().startswith("//")]
        tests = [i for i, ln in enumerate(lines) if ln.startswith("s_cbranch_vccnz") and i >= 1 and
                 any(l2.startswith("v_cmp_ge_u32_e32 vcc, s") for l2 in lines[max(0, i - 3):i])]
        if len(tests) < 2:
            return None
        seg = ops[tests[0] + 1:tests[-1] + 1]
        n = len(tests) - 1
        valu = [o for o in seg if o.startswith("v_") and not o.startswith("v_readlane") and not o.startswith("v_readfirstlane")]
        fast = sum(1 for o in valu if re.sub(r"_e(32|64)$", "", o) in FAST)
        lds = sum(1 for o in seg if o.startswith("ds_"))
        return {"positions": n, "valu": round(len(valu) / n, 2), "valu_fast": round(fast / n, 2), "valu_slow": round((len(valu) - fast) / n, 2),
                "lds": round(lds / n, 2), "issue_cycles": round((fast * 2.4 + (len(valu) - fast) * 4.15) / n, 1)}
    return None


def pretty(name):
    m = re.match(r"_ZN2fh9k2_sketchILi(\d+)ELb(\d)ELb(\d)ELb(\d)EEE", name)
    if m:
        return "k2_sketch<%s,%s%s%s>" % (m.group(1), "M" if m.group(2) == "1" else "-", "S0" if m.group(3) == "1" else "--",
                                          "L" if m.group(4) == "1" else "-")
    m = re.match(r"_ZN2fh13k2_sketch_segILi(\d+)ELb(\d)EEE", name)
    if m:
        return "k2_sketch_seg<%s,%s>" % (m.group(1), "S0" if m.group(2) == "1" else "--")
    m = re.match(r"_ZN2fh8k2_batchILi(\d+)ELb(\d)EEE", name)
    if m:
        return "k2_batch<%s,%s>" % (m.group(1), "S0" if m.group(2) == "1" else "--")
    m = re.match(r"_ZN2fh12k2_sketch_wsILi(\d+)EEE", name)
    if m:
        return "k2_sketch_ws<%s>" % m.group(1)
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
    ap.add_argument("--objects", action="store_true", help="read finch_rs_amd/csrc/obj/fh_k2*.o (what libfinch_hip.so was linked from) instead of compiling")
    ap.add_argument("--seg", action="store_true", help="the segment form of the kernel (fh_k2s.hip) instead of fh_k2.hip")
    ap.add_argument("--mix", action="store_true", help="with --k: VALU instruction classes per position of the hot loop (static, from the ISA)")
    args = ap.parse_args()
    kflags = args.flags.split() if args.flags is not None else list(B.K2_FLAGS)
    rows = []
    if args.k is not None and args.mix and not args.asm:
        args.asm = os.path.join(tempfile.mkdtemp(prefix="k2mix_"), "k.s")
    if args.objects:
        import glob
        objs = sorted(glob.glob(os.path.join(CSRC, "obj", "fh_k2_*.o")) + glob.glob(os.path.join(CSRC, "obj", "fh_k2w_*.o")) +
                      glob.glob(os.path.join(CSRC, "obj", "fh_k2s_*.o")) + glob.glob(os.path.join(CSRC, "obj", "fh_k2ws_*.o")) +
                      glob.glob(os.path.join(CSRC, "obj", "fh_k2b_*.o")))
        if len(objs) != 5 * B.NPARTS:
            sys.exit("expected %d sketch-kernel objects under csrc/obj, found %d: build the library first" % (5 * B.NPARTS, len(objs)))
        for o in objs:
            rows += unbundle_object(o)
    elif args.k is not None:
        if args.k <= 32:
            rows = compile_one("fh_k2s.hip" if args.seg else "fh_k2.hip", ["FH_PART=0", "FH_ONLY_K=%d" % args.k] + args.D, kflags, args.asm)
        else:
            rows = compile_one("fh_k2w.hip", ["FH_PART=%d" % ((args.k - 33) // 8)] + args.D, [], args.asm)
            rows = [r for r in rows if "ILi%dE" % args.k in r["name"]]
    else:
        jobs = [("fh_k2.hip", ["FH_PART=%d" % p], kflags) for p in range(B.NPARTS)]
        jobs += [("fh_k2w.hip", ["FH_PART=%d" % p], []) for p in range(B.NPARTS)]
        with ThreadPoolExecutor(max_workers=4) as ex:
            for r in ex.map(lambda j: compile_one(*j), jobs):
                rows += r
    rows = [r for r in rows if "k2_sketch" in r["name"] or "k2_batch" in r["name"]]
    if not args.all_variants:
        rows = [r for r in rows if "k2_sketch_w" in r["name"] or "k2_sketch_seg" in r["name"] or "k2_batch" in r["name"] or "ELb0ELb1ELb0E" in r["name"]]

    def key(r):
        m = re.search(r"ILi(\d+)E", r["name"])
        return (int(m.group(1)) if m else 0, r["name"])
    print("%-28s %5s %6s %8s %5s %6s" % ("kernel", "VGPR", "spill", "scratch", "SGPR", "LDS"))
    for r in sorted(rows, key=key):
        print("%-28s %5s %6s %8s %5s %6s" % (pretty(r["name"]), r["vgpr"], r["spill"], r["scratch"], r["sgpr"], r["lds"]))


    bad = [r for r in rows if r["spill"] not in ("0", "?") or (r["vgpr"].isdigit() and int(r["vgpr"]) > 128)]
    # (the segment kernels keep phase A's loop-invariant addresses in scratch, reloaded once per tile of ~9 000 positions: reported,
    # not failed on -- `--seg --mix` counts the scratch instructions inside the positions' code, which must be none)
    bad = [r for r in bad if "k2_sketch_seg" not in r["name"] or (r["vgpr"].isdigit() and int(r["vgpr"]) > 128)]
    if args.mix and args.k is not None:
        sub = ("k2_sketch_segILi%dE" % args.k if args.seg else "k2_sketchILi%dELb0ELb1ELb0E" % args.k) if args.k <= 32 else "k2_sketch_wILi%dE" % args.k
        print("hot loop per position:", hot_mix(args.asm, sub))
    if bad:
        print("FAIL: spilled registers or more than 128 VGPRs:", ", ".join(pretty(r["name"]) for r in bad))
        sys.exit(1)


if __name__ == "__main__":
    main()
