#!/usr/bin/env python3
"""Derive the per-wave-iteration figures bench.py reports under roofline.pmc / extras.*.pmc from a rocprofv3 PMC set.

    python tools/pmc_summary.py <key> <profiles/rNN_pmc_k2_*.json> <positions> [<launches>]

<key>       workload key bench.py looks up (c2_k21_n1000, c2_k31_n1000, ...)
<positions> k-mer start positions the profiled k2_sketch launches covered (bench.py's roofline.alg_bytes_per_launch x
            launches of the `--steps 1 --warmup 0` run the counters were collected over)
Writes / updates profiles/pmc_summary.json.  Conventions (profiles/README.md): one wave-iteration = 64 positions, one per
lane; 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE is in KB and doubled per the guide's gfx950
correction for 16 B/lane streaming reads, WRITE_SIZE in KB uncorrected.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, XCDS = 1024, 8


def main():
    key, path, positions = sys.argv[1], sys.argv[2], float(sys.argv[3])
    d = json.load(open(path))

    def c(name):
        v = d.get(name)
        return float(v["sum"]) if isinstance(v, dict) else (float(v) if v is not None else None)
    outp = None
    if "--out" in sys.argv:
        i = sys.argv.index("--out")
        outp = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    launches = int(sys.argv[4]) if len(sys.argv) > 4 else int(d["SQ_INSTS_VALU"]["dispatches"])
    wi = positions / 64.0
    cycles = c("GRBM_GUI_ACTIVE") / XCDS  # device cycles while the launches ran
    out = {
        "source": os.path.relpath(path, ROOT),
        "positions": int(positions), "launches": launches,
        "valu_per_wave_iter": round(c("SQ_INSTS_VALU") / wi, 2),
        "salu_per_wave_iter": round(c("SQ_INSTS_SALU") / wi, 2) if c("SQ_INSTS_SALU") else None,
        "lds_insts_per_wave_iter": round(c("SQ_INSTS_LDS") / wi, 2) if c("SQ_INSTS_LDS") else None,
        "lds_active_per_wave_iter": round(c("SQ_LDS_IDX_ACTIVE") / wi, 2) if c("SQ_LDS_IDX_ACTIVE") else None,
        "lds_bank_conflict_per_wave_iter": round(c("SQ_LDS_BANK_CONFLICT") / wi, 2) if c("SQ_LDS_BANK_CONFLICT") else None,
        # cycles one SIMD has per wave-iteration it retires, and per VALU instruction: the issue limit is ~4 (2.4 for the
        # plain 32-bit and/or/xor/add/sub/shift-right, profiles/r01_ubench_valu_rates.txt)
        "cycles_per_wave_iter": round(cycles * SIMDS / wi, 1),
        "cycles_per_valu_inst": round(cycles * SIMDS / c("SQ_INSTS_VALU"), 3),
        "wait_any_frac_of_wave_cycles": round(c("SQ_WAIT_ANY") / c("SQ_WAVE_CYCLES"), 3) if c("SQ_WAIT_ANY") and c("SQ_WAVE_CYCLES") else None,
    }
    # The issue-cycle model of the hot loop: its VALU instructions per position counted in the ISA (tools/k2_regs.py --mix),
    # priced at what gfx950 issues them at (~2.4 cycles the plain and/or/xor/add/shift-right/mov, ~4.15 everything else:
    # profiles/r01_ubench_valu_rates.txt) -- the share of the measured cycles per wave-iteration that is pure VALU issue.
    try:
        import re
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import k2_regs
        K = int(re.search(r"_k(\d+)_", key).group(1))
        if K <= 32:
            asm = os.path.join(tempfile.mkdtemp(prefix="pmcmix_"), "k.s")
            sys.path.insert(0, k2_regs.CSRC)
            import importlib.util
            spec = importlib.util.spec_from_file_location("fh_build", os.path.join(k2_regs.CSRC, "build.py"))
            B = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(B)
            seg = any("k2_sketch_seg" in n for n in (d.get("kernel_names") or []))  # which form of the kernel the counters are of
            out["kernel"] = ("k2_sketch_seg<%d>" if seg else "k2_sketch<%d>") % K
            k2_regs.compile_one("fh_k2s.hip" if seg else "fh_k2.hip", ["FH_PART=0", "FH_ONLY_K=%d" % K], list(B.K2_FLAGS), asm)
            mix = k2_regs.hot_mix(asm, ("k2_sketch_segILi%dE" if seg else "k2_sketchILi%dELb0ELb1ELb0E") % K)
            if mix:
                out["valu_issue_model"] = {"hot_loop_valu_fast_per_position": mix["valu_fast"], "hot_loop_valu_slow_per_position": mix["valu_slow"],
                                           "hot_loop_lds_per_position": mix["lds"], "issue_cycles_per_wave_iter": mix["issue_cycles"],
                                           "frac_of_measured_cycles": round(mix["issue_cycles"] / out["cycles_per_wave_iter"], 3)}
    except Exception as e:  # noqa: BLE001 -- no compiler here: the counters stand on their own
        out["valu_issue_model"] = {"error": str(e)[:200]}
    if c("FETCH_SIZE") is not None:
        hbm = 2.0 * c("FETCH_SIZE") * 1024.0 + (c("WRITE_SIZE") or 0.0) * 1024.0
        out["hbm_bytes_per_launch"] = int(hbm / launches)
        out["hbm_bytes_per_position"] = round(hbm / positions, 4)
    p = outp or os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        allp = json.load(open(p))
    except Exception:
        allp = {}
    allp[key] = out
    json.dump(allp, open(p, "w"), indent=1, sort_keys=True)
    print(key, json.dumps(out))


if __name__ == "__main__":
    main()
