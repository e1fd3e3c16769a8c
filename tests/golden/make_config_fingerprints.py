#!/usr/bin/env python3
"""Generate tests/golden/config_fingerprints.json: the fingerprint of the sketch the REFERENCE ALGORITHM gives for
BASELINE.json's configurations, computed on the CPU from the oracle alone (no GPU, no libfinch_hip kernel).

    python tests/golden/make_config_fingerprints.py [--only c2_k21_n1000,...] [--procs P]

How: the synthetic read set of SURVEY.md 8d M4 (host generator, seed 20250620) is cut into contiguous read blocks; every block is
sketched by the oracle (oracle/finch_oracle.c = mash.rs:34-63 restated) in its own process; the block sketches are merged with the
size-independent property of SURVEY 8e (global bottom-n = bottom-n of the union, counts summed, the k-mer of the first block in
stream order) in numpy -- the merge of tests/test_gpu_full_size.py, not the product's.  bench.py prints the same seven numbers
for the sketch the GPUs produced ("sketch_check") and exits non-zero if they differ from this file.

The fingerprint: n_hashes, min_hash, max_hash, hash_xor (xor of all hashes), count_sum, extra_sum, kmer_byte_sum (sum of all
k-mer bytes) -- together with total_kmers (mash.rs:35).  50 Gbase takes ~5 minutes on 8 cores.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED, GL, RL, SUB_PPM, N_PPM = 20250620, 5_000_000, 150, 10_000, 500

CONFIGS = {
    # name: (gbases, k, n)
    "c2_k21_n1000": (10.0, 21, 1000),   # BASELINE configs[1]
    "c4_k21_n1000": (50.0, 21, 1000),   # BASELINE configs[3]
    "c2_k31_n1000": (10.0, 31, 1000),   # configs[1]'s stream at k = 31 (bench.py extras)
}

_G = {}


def _shard(job):
    from finch_rs_amd import sketch_schemes as S
    from oracle import oracle as O
    first, count, k, n = job
    o = O.OracleSketcher(O.MASH, n, k, 0)
    step = 200_000  # reads per generated piece: keeps a worker's footprint at ~30 MB
    for f in range(first, first + count, step):
        c = min(step, first + count - f)
        o.process_packed(S.synth_reads_host(_G["genome"], f, c, RL, SEED, SUB_PPM, N_PPM), 0)
    kc, km = o.to_vec()
    return kc, km, o.total_bases_and_kmers()[1]


def _c5_file(i):
    from finch_rs_amd import sketch_schemes as S
    from oracle import oracle as O
    o = O.OracleSketcher(O.MASH, 1000, 21, 0)
    assert o.sketch_stream(S.synth_fasta_file(i, SEED)) == 1  # the oracle's own FASTA reader (lib.rs:51-94 restated)
    kc, _ = o.to_vec()
    return int(np.bitwise_xor.reduce(kc["hash"])), int(kc["count"].astype(np.uint64).sum()), int(o.total_bases_and_kmers()[1])


def genome_numpy(length, seed):
    """SURVEY 8d M4's genome G restated in numpy from the specification alone -- base i = "ACGT"[splitmix64(splitmix64(seed ^
    "genome") + i) >> 62] -- so that configs[0]'s golden input does not come out of the product's generator"""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)

    def sm(x):
        with np.errstate(over="ignore"):
            x = (x + np.uint64(0x9E3779B97F4A7C15)) & M
            x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
            x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
            return x ^ (x >> np.uint64(31))
    base = sm(np.array([seed ^ 0x67656E6F6D65], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        h = sm(base + np.arange(length, dtype=np.uint64))
    return np.frombuffer(b"ACGT", dtype=np.uint8)[(h >> np.uint64(62)).astype(np.int64)]


def _sm64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _mulhi64(a, b):
    """high 64 bits of a * b (b a Python int < 2^32 or a uint64 array), by 32-bit halves"""
    a = a.astype(np.uint64)
    b = np.uint64(b) if np.isscalar(b) else b.astype(np.uint64)
    m = np.uint64(0xFFFFFFFF)
    s32 = np.uint64(32)
    with np.errstate(over="ignore"):
        al, ah, bl, bh = a & m, a >> s32, b & m, b >> s32
        t = al * bl
        u = ah * bl + (t >> s32)
        v = al * bh + (u & m)
        return ah * bh + (u >> s32) + (v >> s32)


def reads_numpy(genome, first_read, n_reads, read_len, seed, sub_ppm, n_ppm):
    """SURVEY 8d M4's read generator restated in numpy from its specification (the counter-based rules of fh_core.h's
    synth_read_byte, written down independently): read r starts at mulhi(splitmix(h0), |G| - L + 1) with h0 =
    splitmix(splitmix(seed ^ "reads") + r), strand = top bit of h0, per base j a substitution (to one of the three other
    bases) with probability sub_ppm and an 'N' with probability n_ppm from hj = splitmix(h0 + c (j + 1)); one 0 byte behind
    each read.  The goldens of configs[1..3] were generated through the product's host generator; tests/test_oracle_golden.py
    holds the two generators against each other."""
    g = np.asarray(genome, dtype=np.uint8)
    L = read_len
    with np.errstate(over="ignore"):
        base = _sm64(np.array([seed ^ 0x7265616473], dtype=np.uint64))[0]
        h0 = _sm64(base + (np.uint64(first_read) + np.arange(n_reads, dtype=np.uint64)))
        start = _mulhi64(_sm64(h0), len(g) - L + 1).astype(np.int64)
        rev = (h0 >> np.uint64(63)) != 0
        j = np.arange(L, dtype=np.int64)
        idx = np.where(rev[:, None], start[:, None] + (L - 1 - j)[None, :], start[:, None] + j[None, :])
        b = g[idx]
        comp = np.arange(256, dtype=np.uint8)
        comp[ord("A")], comp[ord("C")], comp[ord("G")], comp[ord("T")] = ord("T"), ord("G"), ord("C"), ord("A")
        b = np.where(rev[:, None], comp[b], b)
        hj = _sm64(h0[:, None] + np.uint64(0x632BE59BD9B4E019) * (j[None, :] + 1).astype(np.uint64))
        u_sub = _mulhi64(hj.reshape(-1), 1000000).reshape(hj.shape)
        u_n = _mulhi64(_sm64(hj).reshape(-1), 1000000).reshape(hj.shape)
        code = np.select([b == ord("A"), b == ord("C"), b == ord("G")], [0, 1, 2], 3).astype(np.uint64)
        add = np.uint64(1) + ((((hj >> np.uint64(20)) & np.uint64(0xFFFFF)) * np.uint64(3)) >> np.uint64(20))
        sub = np.frombuffer(b"ACGT", dtype=np.uint8)[((code + add) & np.uint64(3)).astype(np.int64)]
    b = np.where(u_sub < sub_ppm, sub, b)
    b = np.where(u_n < n_ppm, np.uint8(ord("N")), b)
    out = np.zeros((n_reads, L + 1), dtype=np.uint8)
    out[:, :L] = b
    return out.reshape(-1)


def fasta_70(genome, name=b"G"):
    """configs[0]'s input: the genome as ONE record of 70-column lines"""
    n = len(genome)
    full = n // 70 * 70
    body = np.concatenate([genome[:full].reshape(-1, 70), np.full((full // 70, 1), 10, np.uint8)], axis=1).reshape(-1).tobytes()
    tail = genome[full:].tobytes()
    return b">" + name + b"\n" + body + (tail + b"\n" if tail else b"")


def merge_numpy(parts, n):
    kc = np.concatenate([p[0] for p in parts])
    km = np.concatenate([p[1] for p in parts])
    order = np.argsort(kc["hash"], kind="stable")  # stable: block order == stream order within equal hashes
    kc, km = kc[order], km[order]
    uniq, start = np.unique(kc["hash"], return_index=True)
    counts = np.add.reduceat(kc["count"].astype(np.uint64), start)
    extra = np.add.reduceat(kc["extra_count"].astype(np.uint64), start)
    out = np.zeros(len(uniq), dtype=kc.dtype)
    out["hash"] = uniq
    out["count"] = np.minimum(counts, 2**32 - 1)
    out["extra_count"] = np.minimum(extra, 2**32 - 1)
    return out[:n], km[start][:n], sum(p[2] for p in parts)


def fingerprint(kc, km, total_kmers=None):
    fp = {"n_hashes": int(len(kc)), "min_hash": int(kc["hash"][0]) if len(kc) else None,
          "max_hash": int(kc["hash"][-1]) if len(kc) else None,
          "hash_xor": int(np.bitwise_xor.reduce(kc["hash"])) if len(kc) else 0,
          "count_sum": int(kc["count"].astype(np.uint64).sum()),
          "extra_sum": int(kc["extra_count"].astype(np.uint64).sum()),
          "kmer_byte_sum": int(np.asarray(km).astype(np.uint64).sum())}
    if total_kmers is not None:
        fp["total_kmers"] = int(total_kmers)
    return fp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--procs", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(HERE, "config_fingerprints.json"))
    args = ap.parse_args()
    from finch_rs_amd import sketch_schemes as S
    _G["genome"] = S.synth_genome_host(GL, SEED)
    names = args.only.split(",") if args.only else list(CONFIGS)
    try:
        out = json.load(open(args.out))
    except Exception:
        out = {}
    out["_generator"] = ("tests/golden/make_config_fingerprints.py: host read generator (seed %d, genome %d, %d bp, sub %d ppm, "
                         "N %d ppm) -> oracle per read block -> numpy merge" % (SEED, GL, RL, SUB_PPM, N_PPM))
    with mp.get_context("fork").Pool(args.procs) as pool:
        if args.only is None or "c5_files_0_255" in names:
            # BASELINE configs[4]: files 0..255 of the synthetic FASTA batch (bench.py --workload c5), library defaults
            fx = cs = tk = 0
            for a, b, c in pool.map(_c5_file, range(256), chunksize=4):
                fx ^= a; cs += b; tk += c
            out["c5_files_0_255"] = {"sample_files": 256, "hash_xor": fx, "count_sum": cs, "total_kmers": tk, "k": 21, "n": 1000}
            print("c5_files_0_255:", json.dumps(out["c5_files_0_255"]), flush=True)
            json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
        if args.only is None or "c1_fasta_k21_n1000" in names:
            # BASELINE configs[0]: genome G (5 Mb) as a 70-column FASTA file, library defaults (Mash 1000 / 1000, k = 21, seed 0,
            # filters off for FASTA): the oracle's own FASTA reader and sketch_stream (lib.rs:51-94 restated)
            from oracle import oracle as O
            o = O.OracleSketcher(O.MASH, 1000, 21, 0)
            text = fasta_70(genome_numpy(GL, SEED))
            assert o.sketch_stream(text) == 1
            kc, km = o.to_vec()
            tb, tk = o.total_bases_and_kmers()
            fp = fingerprint(kc, km, tk)
            fp.update({"k": 21, "n": 1000, "genome": GL, "seq_length": int(tb), "file_bytes": len(text)})
            out["c1_fasta_k21_n1000"] = fp
            print("c1_fasta_k21_n1000:", json.dumps(fp), flush=True)
            json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
        if args.only is None or "c3_k31_filtered" in names:
            # BASELINE configs[2]: 10 Gbase, k = 31, kmers_to_sketch = 2 000 000 (CLI oversketch x200 of 10 000), then the reference's
            # filters -- strand 0.1, error 0.31 (guess_filter_threshold on what the strand filter left), abundance -- and the cut to
            # final_size 10 000 (filtering.rs:60-87, mod.rs:115-128), all by the oracle
            from oracle import oracle as O
            gb, k, n, final = 10.0, 31, 2_000_000, 10_000
            reads = int(np.ceil(gb * 1e9 / RL))
            blocks = args.procs * 4
            b = np.linspace(0, reads, blocks + 1).astype(np.int64)
            t0 = time.time()
            parts = pool.map(_shard, [(int(b[i]), int(b[i + 1] - b[i]), k, n) for i in range(blocks)], chunksize=1)
            kc, km, tk = merge_numpy(parts, n)
            a, ak = O.filter_strands(kc, km, 0.1)
            cutoff = O.guess_filter_threshold(a, 0.31)
            f, fk = O.filter_abundance(a, ak, cutoff, None)
            f, fk = f[:final], fk[:final]
            fp = fingerprint(f, fk, tk)
            fp.update({"gbases": gb, "reads": reads, "k": k, "kmers_to_sketch": n, "final_size": final, "strand_filter": 0.1, "err_filter": 0.31,
                       "abun_lo": int(cutoff), "oversketch_hash_xor": int(np.bitwise_xor.reduce(kc["hash"])), "read_blocks": blocks})
            out["c3_k31_filtered"] = fp
            print("c3_k31_filtered: %.0f s %s" % (time.time() - t0, json.dumps(fp)), flush=True)
            json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
        for name in [n for n in names if n in CONFIGS]:
            gb, k, n = CONFIGS[name]
            reads = int(np.ceil(gb * 1e9 / RL))
            blocks = args.procs * 8
            b = np.linspace(0, reads, blocks + 1).astype(np.int64)
            t0 = time.time()
            parts = pool.map(_shard, [(int(b[i]), int(b[i + 1] - b[i]), k, n) for i in range(blocks)], chunksize=1)
            kc, km, tk = merge_numpy(parts, n)
            fp = fingerprint(kc, km, tk)
            fp.update({"gbases": gb, "reads": reads, "k": k, "n": n, "read_blocks": blocks})
            out[name] = fp
            print("%s: %.0f s %s" % (name, time.time() - t0, json.dumps(fp)), flush=True)
            json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
