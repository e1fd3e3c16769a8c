"""Full-size parity (BASELINE.json configs[1]: 10 Gbase of synthetic 150 bp reads, k=21, n=1000).

The GPU sketches the whole stream resident in HBM (the bench path).  The oracle cannot do 10 Gbase on one
core in test time, so it runs on read-block shards in a process pool and the shard sketches are merged
with the size-independent property of SURVEY 8e (global bottom-n = bottom-n of the union of shard
sketches, counts summed) -- implemented here in numpy, independently of the product's merge code.
Bit-exact comparison of hashes, counts, extra_counts and k-mer bytes.  Size via FH_FULL_GBASES (default 10)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SEED, GL, RL = 20250620, 5_000_000, 150
K, N = 21, 1000


def _oracle_shard(args):
    genome, first, count = args[:3]
    k, n = (args[3], args[4]) if len(args) > 3 else (K, N)
    reads = S.synth_reads_host(genome, first, count, RL, SEED, 10000, 500)  # same generator as the device (tested equal)
    o = O.OracleSketcher(O.MASH, n, k, 0)
    o.process_packed(reads, 0)
    kc, km = o.to_vec()
    return kc, km, o.total_bases_and_kmers()[1]


def merge_numpy(parts, n):
    kc = np.concatenate([p[0] for p in parts])
    km = np.concatenate([p[1] for p in parts])
    order = np.argsort(kc["hash"], kind="stable")  # stable: shard order == stream order within equal hashes
    kc, km = kc[order], km[order]
    uniq, start = np.unique(kc["hash"], return_index=True)
    counts = np.add.reduceat(kc["count"].astype(np.uint64), start)
    extra = np.add.reduceat(kc["extra_count"].astype(np.uint64), start)
    out = np.zeros(len(uniq), dtype=kc.dtype)
    out["hash"] = uniq
    out["count"] = np.minimum(counts, 2**32 - 1)
    out["extra_count"] = np.minimum(extra, 2**32 - 1)
    return out[:n], km[start][:n], sum(p[2] for p in parts)


def test_full_size_stream_bit_exact_vs_sharded_oracle():
    gbases = float(os.environ.get("FH_FULL_GBASES", "10"))
    n_reads = int(np.ceil(gbases * 1e9 / RL))
    rec = RL + 1
    ncpu = max(1, min(len(os.sched_getaffinity(0)), 96))  # more processes than granted cores only cost a little
    if ncpu < 16 and "FH_FULL_GBASES" not in os.environ:
        gbases = 1.0  # keep the CPU side of the check within a minute on small hosts
        n_reads = int(np.ceil(gbases * 1e9 / RL))
    # --- GPU: whole stream resident, one sketcher (the bench path) ---
    dg = F.DeviceBuffer(GL)
    dr = F.DeviceBuffer(n_reads * rec + 64)
    S.synth_genome_device(dg, GL, SEED)
    S.synth_reads_device(dr, dg, GL, 0, n_reads, RL, SEED, 10000, 500)
    sk = F.SketchParams.mash(N, N, True, K, 0).create_sketcher()
    sk.push_device(dr.ptr, n_reads * rec)
    kc, km, _ = sk.to_arrays()
    tk = sk.finish()[1]
    # spot-check that what sits in HBM is what the host generator makes (full equality is a separate test)
    genome = S.synth_genome_host(GL, SEED)
    for first in (0, n_reads // 2, n_reads - 1000):
        host = S.synth_reads_host(genome, first, 1000, RL, SEED, 10000, 500)
        assert np.array_equal(dr.download(1000 * rec, first * rec), host)
    # --- oracle on shards, merged ---
    shards = ncpu * 4
    bounds = np.linspace(0, n_reads, shards + 1).astype(np.int64)
    jobs = [(genome, int(bounds[i]), int(bounds[i + 1] - bounds[i])) for i in range(shards) if bounds[i + 1] > bounds[i]]
    with mp.get_context("fork").Pool(ncpu) as pool:
        parts = pool.map(_oracle_shard, jobs, chunksize=1)
    okc, okm, otk = merge_numpy(parts, N)
    assert len(kc) == N
    assert np.array_equal(kc, okc)
    assert np.array_equal(km, okm)
    assert tk == otk
    # and the product's own sharded path agrees with its single-stream path at this size
    half = (n_reads // 2) // 16 * 16  # device blocks must start 16-byte aligned (151 * half)
    a = F.SketchParams.mash(N, N, True, K, 0).create_sketcher()
    a.push_device(dr.ptr, half * rec)
    a.finish()
    b = F.SketchParams.mash(N, N, True, K, 0).create_sketcher()
    b.set_stream_offset(half * rec)
    b.push_device(dr.ptr + half * rec, (n_reads - half) * rec)
    b.finish()
    a.merge(b)
    m = a.to_arrays()
    assert np.array_equal(m[0], kc) and np.array_equal(m[1], km) and a.finish()[1] == tk


def test_config3_oversketch_and_filtering_vs_sharded_oracle():
    """BASELINE.json configs[2] shape: k=31, final 10 000 hashes, kmers_to_sketch = 2 000 000 (CLI oversketch x200,
    cli.rs:187-192), strand filter 0.1, err filter 1% -> 0.31 (cli.rs:264-265), filtering on the host.
    Default 2 Gbase (FH_FULL_GBASES_C3 to change): device sketch of 2 M hashes bit-exact vs the sharded oracle,
    then filter_counts + process_post_filter through the C++ host layer vs the oracle's filters."""
    from finch_rs_amd import host as H
    gbases = float(os.environ.get("FH_FULL_GBASES_C3", "2"))
    ncpu = max(1, min(len(os.sched_getaffinity(0)), 64))
    if ncpu < 16 and "FH_FULL_GBASES_C3" not in os.environ:
        gbases = 0.2
    k, n_eff, final = 31, 2_000_000, 10_000
    n_reads = int(np.ceil(gbases * 1e9 / RL))
    rec = RL + 1
    dg = F.DeviceBuffer(GL)
    dr = F.DeviceBuffer(n_reads * rec + 64)
    S.synth_genome_device(dg, GL, SEED)
    S.synth_reads_device(dr, dg, GL, 0, n_reads, RL, SEED, 10000, 500)
    params = F.SketchParams.mash(n_eff, final, False, k, 0)
    sk = params.create_sketcher()
    sk.push_device(dr.ptr, n_reads * rec)
    kc, km, _ = sk.to_arrays()
    tk = sk.finish()[1]
    genome = S.synth_genome_host(GL, SEED)
    shards = min(ncpu, 32)
    bounds = np.linspace(0, n_reads, shards + 1).astype(np.int64)
    jobs = [(genome, int(bounds[i]), int(bounds[i + 1] - bounds[i]), k, n_eff) for i in range(shards)]
    with mp.get_context("fork").Pool(shards) as pool:
        parts = pool.map(_oracle_shard, jobs, chunksize=1)
    okc, okm, otk = merge_numpy(parts, n_eff)
    assert np.array_equal(kc, okc) and np.array_equal(km, okm) and tk == otk
    # host filtering (N1) on the device output
    filt = H.FilterParams(True, (None, None), 0.31, 0.1)
    res = H.sketches_from_arrays("c3", n_reads * RL, tk, kc, km, params, H.FilterParams(False))
    fp = res.apply_filters(0, filt)
    got = res.sketch(0)
    a, ak = O.filter_strands(okc, okm, 0.1)
    cutoff = O.guess_filter_threshold(a, 0.31)
    b, bk = O.filter_abundance(a, ak, cutoff, None)
    assert fp.abun_filter == (cutoff, None)
    assert len(got.hashes) == final
    assert np.array_equal(got.arrays[0], b[:final]) and np.array_equal(got.arrays[1], bk[:final])
