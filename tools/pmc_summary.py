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
        # rocprof's VALUBusy: 4 cycles per VALU instruction over the SIMD cycles available
        "valu_busy": round(4.0 * c("SQ_ACTIVE_INST_VALU") / SIMDS / cycles, 3) if c("SQ_ACTIVE_INST_VALU") else None,
        "wait_any_frac_of_wave_cycles": round(c("SQ_WAIT_ANY") / c("SQ_WAVE_CYCLES"), 3) if c("SQ_WAIT_ANY") and c("SQ_WAVE_CYCLES") else None,
    }
    if c("FETCH_SIZE") is not None:
        hbm = 2.0 * c("FETCH_SIZE") * 1024.0 + (c("WRITE_SIZE") or 0.0) * 1024.0
        out["hbm_bytes_per_launch"] = int(hbm / launches)
        out["hbm_bytes_per_position"] = round(hbm / positions, 4)
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        allp = json.load(open(p))
    except Exception:
        allp = {}
    allp[key] = out
    json.dump(allp, open(p, "w"), indent=1, sort_keys=True)
    print(key, json.dumps(out))


if __name__ == "__main__":
    main()
