"""Full-size parity (BASELINE.json configs[1]: 10 Gbase of synthetic 150 bp reads, k=21, n=1000).

The GPU sketches the whole stream resident in HBM (the bench path).  The oracle cannot do 10 Gbase on one
core in test time, so it runs on read-block shards in a process pool and the shard sketches are merged
with the size-independent property of SURVEY 8e (global bottom-n = bottom-n of the union of shard
sketches, counts summed) -- implemented here in numpy, independently of the product's merge code.
Bit-exact comparison of hashes, counts, extra_counts and k-mer bytes.

Every configuration comes as TWO tests, so that the driver's record says which size ran:
  test_*_full    BASELINE.json's size.  Skips -- with the reason -- when the box grants fewer than FULL_MIN_CORES cores for
                 the oracle side (or, C5, the tmpfs does not hold the files): "N passed, 0 skipped" therefore means C2, C3, C4
                 and C5 all ran at BASELINE size.  It never shrinks.
  test_*_scaled  the same check on a cut-down read set (1 / 0.2 / 1 Gbase, C5: 2 % lengths), always runs."""
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
FULL_MIN_CORES = 16  # the oracle side of a full-size check stays within ~2 minutes from here up


def _cores():
    return len(os.sched_getaffinity(0))


FULL_RAN = []  # which BASELINE configurations ran at their full size in this session (conftest.py prints it)


def _cannot_run_full(why):
    """a box too small for a full-size check: skip with the reason -- unless FH_REQUIRE_FULL=1 says that the full sizes are the
    point of the run (tools/gpu_evidence.sh suite), where a silent fall-back to the _scaled twin must be a FAILURE"""
    if os.environ.get("FH_REQUIRE_FULL"):
        pytest.fail("FH_REQUIRE_FULL=1: " + why)
    pytest.skip(why)


def _need_cores_for_full():
    if _cores() < FULL_MIN_CORES and not os.environ.get("FH_FORCE_FULL"):
        _cannot_run_full("full BASELINE size needs >= %d cores for the oracle side, %d granted (FH_FORCE_FULL=1 to run anyway); the "
                    "_scaled twin of this test ran" % (FULL_MIN_CORES, _cores()))


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


def test_c2_10gbase_stream_bit_exact_vs_sharded_oracle_full():
    """BASELINE.json configs[1] at its size: 10 Gbase"""
    _need_cores_for_full()
    _c2_stream(10.0)
    FULL_RAN.append("configs[1] (10 Gbase)")


def test_c2_stream_bit_exact_vs_sharded_oracle_scaled():
    _c2_stream(1.0)


def _c2_stream(gbases):
    n_reads = int(np.ceil(gbases * 1e9 / RL))
    rec = RL + 1
    ncpu = max(1, min(_cores(), 96))  # more processes than granted cores only cost a little
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


def test_c3_10gbase_oversketch_and_filtering_vs_sharded_oracle_full():
    """BASELINE.json configs[2] at its size: 10 Gbase, k=31, 2 M hashes, filters"""
    _need_cores_for_full()
    _c3_oversketch_and_filtering(10.0)
    FULL_RAN.append("configs[2] (10 Gbase, k=31, 2 M hashes, filters)")


def test_c3_oversketch_and_filtering_vs_sharded_oracle_scaled():
    _c3_oversketch_and_filtering(0.2)


def _c3_oversketch_and_filtering(gbases):
    """BASELINE.json configs[2] shape: k=31, final 10 000 hashes, kmers_to_sketch = 2 000 000 (CLI oversketch x200,
    cli.rs:187-192), strand filter 0.1, err filter 1% -> 0.31 (cli.rs:264-265), filtering on the host:
    device sketch of 2 M hashes bit-exact vs the sharded oracle, then filter_counts + process_post_filter through the C++
    host layer vs the oracle's filters."""
    from finch_rs_amd import host as H
    ncpu = max(1, min(_cores(), 64))
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
    # the same straight off the sketcher (the tail of sketch_stream, lib.rs:70-93, for a caller that fed the device itself)
    direct = H.sketch_from_sketcher(sk, "c3", n_reads * RL, 2, params, H.FilterParams(None, (None, None), 0.31, 0.1)).sketch(0)
    assert direct.filter_params.filter_on is True and direct.filter_params.abun_filter == (cutoff, None)  # FASTQ: filtering on by default
    assert np.array_equal(direct.arrays[0], b[:final]) and np.array_equal(direct.arrays[1], bk[:final])
    assert (direct.seq_length, direct.num_valid_kmers) == (n_reads * RL, tk)
    if gbases == 10.0:  # ... and that is the sketch bench.py's self-check of configs[2] expects
        import json
        g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_fingerprints.json")))["c3_k31_filtered"]
        d0 = direct.arrays[0]
        assert (g["hash_xor"], g["count_sum"], g["extra_sum"], g["total_kmers"], g["abun_lo"]) == (
            int(np.bitwise_xor.reduce(d0["hash"])), int(d0["count"].astype(np.uint64).sum()), int(d0["extra_count"].astype(np.uint64).sum()), tk, cutoff)


def test_c4_50gbase_sharded_read_blocks_and_host_merge_full():
    """BASELINE.json configs[3] at its size: 50 Gbase (and the result must also carry the committed golden fingerprint)"""
    _need_cores_for_full()
    _c4_sharded_read_blocks_and_host_merge(50.0)
    FULL_RAN.append("configs[3] (50 Gbase, read blocks + merge)")


def test_c4_sharded_read_blocks_and_host_merge_scaled():
    _c4_sharded_read_blocks_and_host_merge(1.0)


def _c4_sharded_read_blocks_and_host_merge(gbases):
    """BASELINE.json configs[3] on ONE MI355X: the stream (50.3 GB at full size) is resident, sketched
    (a) as one stream and (b) as the 8 read blocks the 8 ranks of a node would take (shard_bounds, one handle per block,
    fh_set_stream_offset, then the host merge of the 8 partial sketches in their wire format, fh_merge_wire -- the
    bench.py --gpus 8 path minus the transport).  Both must equal the oracle run on 256 read-block shards and merged
    with the independent numpy merge above (the size-independent property of SURVEY 8e)."""
    from finch_rs_amd import sharding as SH
    ncpu = max(1, min(_cores(), 96))
    n_reads = int(np.ceil(gbases * 1e9 / RL))
    rec = RL + 1
    dg = F.DeviceBuffer(GL)
    dr = F.DeviceBuffer(n_reads * rec + 64)
    S.synth_genome_device(dg, GL, SEED)
    S.synth_reads_device(dr, dg, GL, 0, n_reads, RL, SEED, 10000, 500)
    params = F.SketchParams.mash(N, N, True, K, 0)
    # (a) one stream
    sk = params.create_sketcher()
    sk.push_device(dr.ptr, n_reads * rec)
    kc, km, pos = sk.to_arrays()
    tk = sk.finish()[1]
    assert len(kc) == N
    # (b) 8 read blocks, each on its own handle at its own stream offset; partial sketches through the wire format
    world = 8
    bufs = []
    for r in range(world):
        lo, hi = SH.shard_bounds(n_reads, r, world)
        lo, hi = (lo // 16) * 16, (hi // 16) * 16 if r + 1 < world else hi  # device blocks start 16-byte aligned
        h = params.create_sketcher()
        h.set_stream_offset(lo * rec)
        h.push_device(dr.ptr + lo * rec, (hi - lo) * rec)
        pkc, pkm, ppos = h.to_arrays()
        bufs.append(SH.pack_partial(pkc, pkm, ppos, h.finish()[1], N, K))
        h.close()
    mkc, mkm, mpos, mtk = SH.merge_wire(params, bufs, N)
    assert np.array_equal(mkc, kc) and np.array_equal(mkm, km) and np.array_equal(mpos, pos) and mtk == tk
    # oracle on 256 shards (a shard's reads are generated inside its worker: ~200 MB each)
    genome = S.synth_genome_host(GL, SEED)
    for first in (0, n_reads // 3, n_reads - 1000):
        assert np.array_equal(dr.download(1000 * rec, first * rec), S.synth_reads_host(genome, first, 1000, RL, SEED, 10000, 500))
    shards = 256
    bounds = np.linspace(0, n_reads, shards + 1).astype(np.int64)
    jobs = [(genome, int(bounds[i]), int(bounds[i + 1] - bounds[i])) for i in range(shards) if bounds[i + 1] > bounds[i]]
    with mp.get_context("fork").Pool(ncpu) as pool:
        parts = pool.map(_oracle_shard, jobs, chunksize=1)
    okc, okm, otk = merge_numpy(parts, N)
    assert np.array_equal(kc, okc) and np.array_equal(km, okm) and tk == otk
    if gbases == 50.0:  # ... and that is the sketch bench.py's self-check expects
        import json
        g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_fingerprints.json")))["c4_k21_n1000"]
        assert (g["hash_xor"], g["count_sum"], g["extra_sum"], g["total_kmers"]) == (
            int(np.bitwise_xor.reduce(kc["hash"])), int(kc["count"].sum()), int(kc["extra_count"].sum()), tk)


_C5 = {"scale": 1.0}  # inherited by the fork()ed pool workers


def _c5_write(args):
    d, i = args
    with open(os.path.join(d, "g%05d.fa" % i), "wb") as f:
        f.write(S.synth_fasta_file(i, SEED, _C5["scale"]))


def _c5_oracle(i):
    o = O.OracleSketcher(O.MASH, N, K, 0)
    assert o.sketch_stream(S.synth_fasta_file(i, SEED, _C5["scale"])) == 1
    return o.to_vec() + (o.total_bases_and_kmers(),)


def _c5_room():
    import shutil
    import tempfile
    cands = [d for d in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(d)]
    base = max(cands, key=lambda d: shutil.disk_usage(d).free)
    free = shutil.disk_usage(base).free
    if base == "/dev/shm":  # tmpfs pages are RAM: leave room for the processes
        import psutil
        free = min(free, psutil.virtual_memory().available - (24 << 30))
    return base, free


def test_c5_batch_of_10k_fastas_through_sketch_files_full():
    """BASELINE.json configs[4] at its size: 10 000 files of log-uniform 1-10 Mb (~39 GB of text), scale == 1.0"""
    _need_cores_for_full()
    base, free = _c5_room()
    need = 3.95e6 * 1.015 * 10000 / 0.8  # mean of the log-uniform lengths + newlines, with headroom
    if free < need and not os.environ.get("FH_FORCE_FULL"):
        _cannot_run_full("full BASELINE size needs %.0f GB under %s, %.0f GB free; the _scaled twin of this test ran" % (need / 1e9, base, free / 1e9))
    _c5_batch(10000, 1.0, base)
    FULL_RAN.append("configs[4] (10 000 FASTA files)")


def test_c5_batch_of_fastas_through_sketch_files_scaled():
    _c5_batch(10000, 0.02, _c5_room()[0])


def _c5_batch(n_files, scale, base):
    """BASELINE.json configs[4] on one GPU: synthetic RefSeq-sized FASTAs (log-uniform 1-10 Mb x scale, 70-column lines)
    through ONE finch_sketch_files call (lib.rs:29-49): one sketch per file in input order with the file's name, seq_length
    and n = 1000 hashes; a seeded sample of 256 files is compared bit-exact (hashes, counts, k-mers, seq_length,
    numValidKmers) with the oracle's own sketch_stream on the same bytes."""
    import shutil
    import tempfile
    from finch_rs_amd import host as H
    ncpu = max(1, min(_cores(), 64))
    _C5["scale"] = scale
    d = tempfile.mkdtemp(prefix="finch_c5_", dir=base)
    try:
        with mp.get_context("fork").Pool(ncpu) as pool:
            pool.map(_c5_write, [(d, i) for i in range(n_files)], chunksize=16)
            paths = [os.path.join(d, "g%05d.fa" % i) for i in range(n_files)]
            res = H.sketch_files(paths, F.SketchParams.default(), H.FilterParams(None))
            assert len(res) == n_files
            L = H.lib()
            for i in range(n_files):
                ln = S.synth_fasta_length(i, SEED, scale)
                assert L.finch_sketch_name(res._p, i).decode() == paths[i]
                assert L.finch_sketch_seq_length(res._p, i) == ln + (ln - 1) // 70  # raw region: bases + inner newlines
                assert L.finch_sketch_n_hashes(res._p, i) == N
            rng = np.random.default_rng(SEED)
            sample = sorted(rng.choice(n_files, size=min(256, n_files), replace=False).tolist())
            oracles = pool.map(_c5_oracle, sample, chunksize=4)
        for i, (okc, okm, totals) in zip(sample, oracles):
            sk = res.sketch(i)
            assert np.array_equal(sk.arrays[0], okc) and np.array_equal(sk.arrays[1], okm), i
            assert (sk.seq_length, sk.num_valid_kmers) == totals, i
            assert sk.filter_params.filter_on is False  # lib.rs:70-76: FASTA defaults to no filtering
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_c1_ecoli_sized_fasta_through_sketch_files(tmp_path, golden_dir):
    """BASELINE.json configs[0] by name: genome G -- 5 Mb, here from the numpy restatement of the generator's
    specification in tests/golden/make_config_fingerprints.py, not from the product's -- written as ONE record of 70-column
    lines, through finch_sketch_files with the library defaults (Mash 1000 / 1000, k = 21, seed 0; FASTA: no filtering).
    Bit-exact against the oracle's own sketch_stream on the same bytes, and against the committed golden fingerprint
    c1_fasta_k21_n1000 -- through the host-packed small-file path and through the device-side FASTA splitter."""
    import json
    import sys
    from finch_rs_amd import host as H
    sys.path.insert(0, golden_dir)
    import make_config_fingerprints as M
    text = M.fasta_70(M.genome_numpy(M.GL, M.SEED))
    path = str(tmp_path / "G.fa")
    with open(path, "wb") as f:
        f.write(text)
    ora = O.OracleSketcher(O.MASH, 1000, 21, 0)
    assert ora.sketch_stream(text) == 1
    okc, okm = ora.to_vec()
    golden = json.load(open(os.path.join(golden_dir, "config_fingerprints.json")))["c1_fasta_k21_n1000"]
    for small_host in ("1", "0"):
        F.debug_set(small_fasta_host=small_host)
        res = H.sketch_files([path], F.SketchParams.default(), H.FilterParams(None))
        sk = res.sketch(0)
        assert np.array_equal(sk.arrays[0], okc) and np.array_equal(sk.arrays[1], okm)
        assert (sk.seq_length, sk.num_valid_kmers) == ora.total_bases_and_kmers() == (golden["seq_length"], golden["total_kmers"])
        fp = M.fingerprint(sk.arrays[0], sk.arrays[1], sk.num_valid_kmers)
        assert all(fp[key] == golden[key] for key in fp), (fp, golden)
        assert sk.filter_params.filter_on is False
    F.debug_set(small_fasta_host=None)
    # the same file through the device-side splitter, whatever this process read from the environment first
    import subprocess
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); import finch_rs_amd as F; from finch_rs_amd import host as H; "
            "sk = H.sketch_files([%r], F.SketchParams.default(), H.FilterParams(None)).sketch(0); "
            "print(json.dumps([int(np.bitwise_xor.reduce(sk.arrays[0]['hash'])), int(sk.seq_length), int(sk.num_valid_kmers)]))"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path))
    r = subprocess.run([sys.executable, "-c", code], env=F.debug_env(small_fasta_host="0"), stdout=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0
    assert json.loads(r.stdout.strip().splitlines()[-1]) == [golden["hash_xor"], golden["seq_length"], golden["total_kmers"]]
