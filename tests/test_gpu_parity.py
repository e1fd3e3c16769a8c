"""Parity of the HIP path (through the C ABI) against the CPU oracle: bit-exact hashes, counts,
extra_counts and retained k-mer bytes.  Needs a real MI355X: run with `-m gpu`."""
import json
import os

import numpy as np
import pytest

import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _kmers(km):
    return [bytes(r).decode() for r in km]


def assert_same(hip: F.HipSketcher, ora: O.OracleSketcher, ctx=""):
    kc, km, _ = hip.to_arrays()
    okc, okm = ora.to_vec()
    assert len(kc) == len(okc), (ctx, len(kc), len(okc))
    assert np.array_equal(kc["hash"], okc["hash"]), ctx
    assert np.array_equal(kc["count"], okc["count"]), ctx
    assert np.array_equal(kc["extra_count"], okc["extra_count"]), ctx
    assert np.array_equal(km, okm), ctx
    assert hip.finish()[1] == ora.total_bases_and_kmers()[1], ctx


def random_reads(rng, n_reads, lo=0, hi=200, p_n=0.01, p_lower=0.05, genome=None):
    out = []
    for _ in range(n_reads):
        L = int(rng.integers(lo, hi + 1))
        if genome is not None and L > 0:
            st = int(rng.integers(0, len(genome) - L))
            r = genome[st:st + L].copy()
        else:
            r = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L)
        m = rng.random(L)
        r[m < p_n] = ord("N")
        low = (m > 1 - p_lower)
        r[low] = r[low] | 0x20
        out.append(bytes(r))
    return out


@pytest.fixture(scope="module")
def vec(golden_dir):
    with open(os.path.join(golden_dir, "reference_vectors.json")) as f:
        return json.load(f)


def fasta_records(data: bytes):
    """raw sequence() slices of a FASTA file as needletail yields them (newlines inside)"""
    recs = []
    for chunk in data.split(b">")[1:]:
        nl = chunk.index(b"\n")
        recs.append(chunk[nl + 1:].rstrip(b"\r\n"))
    return recs


@pytest.mark.parametrize("kind", ["mash", "scaled"])
def test_cli_golden_kmers(vec, golden_dir, kind):
    # cli/tests/test_cli.rs:80-149
    v = vec["test_cli_rs_99_143"]
    data = open(os.path.join(golden_dir, v["file"]), "rb").read()
    params = (F.SketchParams.mash(v["n"], v["n"], False, v["k"], v["seed"]) if kind == "mash"
              else F.SketchParams.scaled(v["n"], v["k"], v["scale"], v["seed"]))
    sk = params.create_sketcher()
    for rec in fasta_records(data):
        sk.process(rec)
    kc, km, _ = sk.to_arrays()
    assert _kmers(km)[:10] == v["kmers"]
    assert [int(h) for h in kc["hash"][:10]] == [
        933085113509804, 8582128962097342, 12581283643378369, 13388215406653903, 59671498055219043,
        85163822212241463, 196329111101504065, 240583695071237384, 241465901919730030, 256930375650047524]
    assert [int(c) for c in kc["count"][:10]] == [1, 1, 1, 1, 1, 1, 2, 1, 1, 2]
    assert [int(c) for c in kc["extra_count"][:10]] == [0, 0, 1, 1, 1, 1, 2, 1, 1, 0]
    assert sk.total_bases_and_kmers() == (134 + 136 + 135, 339)


def test_longer_sequence_hashes(vec):
    # mash.rs:136-154
    v = vec["mash_rs_141_153"]
    sk = F.SketchParams.mash(100, 100, True, v["k"], v["seed"]).create_sketcher()
    sk.process(v["sequence"].encode())
    kc, _, _ = sk.to_arrays()
    assert [str(int(h)) for h in kc["hash"]] == v["hashes"]


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7, 8, 11, 15, 16, 17, 20, 21, 24, 27, 31, 32,
                               33, 34, 40, 47, 48, 49, 56, 57, 62, 63, 64])  # > 32: two-word k-mers (fh_k2w.hip)
def test_random_reads_all_k(k):
    rng = np.random.default_rng(1000 + k)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=20000)
    reads = random_reads(rng, 400, 0, 180, genome=genome)
    seed = 0 if k % 2 else 42
    n = 64
    sk = F.SketchParams.mash(n, n, True, k, seed).create_sketcher()
    ora = O.OracleSketcher(O.MASH, n, k, seed)
    for r in reads:
        sk.process(r)
        ora.process(r)
    assert_same(sk, ora, "k=%d" % k)
    assert sk.total_bases == ora.total_bases_and_kmers()[0]


@pytest.mark.parametrize("n", [0, 1, 10, 1000, 3000])
def test_sizes(n):
    rng = np.random.default_rng(5 + n)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=300000)
    reads = random_reads(rng, 3000, 30, 160, genome=genome)
    block = b"".join(r + b"\x00" for r in reads)
    sk = F.SketchParams.mash(n, n, True, 21, 0).create_sketcher()
    sk.push_block(block)
    ora = O.OracleSketcher(O.MASH, n, 21, 0)
    ora.process_packed(block, 0)
    assert_same(sk, ora, "n=%d" % n)


def test_fewer_distinct_than_n_and_empty_inputs():
    sk = F.SketchParams.mash(1000, 1000, True, 21, 0).create_sketcher()
    ora = O.OracleSketcher(O.MASH, 1000, 21, 0)
    for r in [b"", b"ACGT", b"N" * 50, b"ACGTACGTACGTACGTACGTACGTACGT" * 3, b"acgtacgtacgtacgtacgtaNgtacgtacgtacgtacgtacgtacgtacgtttt"]:
        sk.process(r)
        ora.process(r)
    assert_same(sk, ora)
    sk2 = F.SketchParams.mash(10, 10, True, 21, 0).create_sketcher()
    assert sk2.to_vec() == [] and sk2.total_bases_and_kmers() == (0, 0)


def test_whitespace_skipped_and_breakers():
    a = F.SketchParams.mash(100, 100, True, 5, 0).create_sketcher()
    b = O.OracleSketcher(O.MASH, 100, 5, 0)
    for r in [b"ACGTT\nGCAAT\r\nCCGA", b"AC GT\tTGCA-ATC.CGA~TTGACA", b"ACGUUGCAuuACGRYACGTAC"]:
        a.process(r)
        b.process(r)
    assert_same(a, b)


@pytest.mark.parametrize("size,scale", [(0, 0.01), (50, 0.001), (3, 1.0), (200, 0.05)])
def test_scaled(size, scale):
    rng = np.random.default_rng(77)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=50000)
    reads = random_reads(rng, 600, 30, 160, genome=genome)
    k = 15
    sk = F.SketchParams.scaled(size, k, scale, 0).create_sketcher()
    ora = O.OracleSketcher(O.SCALED, size, k, 0, scale)
    for r in reads:
        sk.process(r)
        ora.process(r)
    assert_same(sk, ora, "scaled %d %g" % (size, scale))


@pytest.mark.parametrize("n,k", [(3001, 21), (20000, 31), (250000, 21)])
def test_large_sketch_sizes_device_wide_selection(n, k):
    """kmers_to_sketch beyond the in-LDS selection (the CLI's oversketch x200 regime, cli.rs:187-192)"""
    gl, nr, rl, seed = 400000, 60000, 150, 11
    g = S.synth_genome_host(gl, seed)
    reads = S.synth_reads_host(g, 0, nr, rl, seed, 10000, 500)
    sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher(max_launch=1 << 20)
    sk.push_block(reads)
    ora = O.OracleSketcher(O.MASH, n, k, 0)
    ora.process_packed(reads, 0)
    assert_same(sk, ora, "n=%d" % n)


def test_scaled_unbounded_growth():
    """scale=1 keeps every distinct k-mer: the device table has to grow with the input (scaled.rs:118-138)"""
    gl, nr, rl, seed = 200000, 8000, 150, 5
    g = S.synth_genome_host(gl, seed)
    reads = S.synth_reads_host(g, 0, nr, rl, seed, 10000, 500)
    for size, scale in [(3, 1.0), (100, 0.25)]:
        sk = F.SketchParams.scaled(size, 21, scale, 0).create_sketcher(max_launch=8192)
        sk.push_block(reads)
        ora = O.OracleSketcher(O.SCALED, size, 21, 0, scale)
        ora.process_packed(reads, 0)
        assert_same(sk, ora, "scaled growth %g" % scale)
        assert len(sk.to_arrays()[0]) > 100000 * scale


def test_synth_generator_host_equals_device():
    gl, nr, rl, seed = 100000, 5000, 150, 20250620
    g_host = S.synth_genome_host(gl, seed)
    r_host = S.synth_reads_host(g_host, 17, nr, rl, seed, 10000, 500)
    dg = F.DeviceBuffer(gl)
    dr = F.DeviceBuffer(nr * (rl + 1))
    S.synth_genome_device(dg, gl, seed)
    S.synth_reads_device(dr, dg, gl, 17, nr, rl, seed, 10000, 500)
    assert np.array_equal(dg.download(gl), g_host)
    assert np.array_equal(dr.download(nr * (rl + 1)), r_host)
    assert set(np.unique(r_host)) <= set(b"ACGTN\x00")
    assert 0.0002 < np.mean(r_host == ord("N")) < 0.001


@pytest.mark.parametrize("k,n", [(21, 1000), (31, 2000)])
def test_resident_stream_vs_oracle(k, n):
    """the bench path: synthetic 150 bp reads generated in HBM, sketched with fh_push_device"""
    gl, nr, rl, seed = 500000, 200000, 150, 20250620
    dg = F.DeviceBuffer(gl)
    nbytes = nr * (rl + 1)
    dr = F.DeviceBuffer(nbytes + 64)
    S.synth_genome_device(dg, gl, seed)
    S.synth_reads_device(dr, dg, gl, 0, nr, rl, seed, 10000, 500)
    sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher(max_launch=4 << 20)
    sk.push_device(dr.ptr, nbytes)
    host = dr.download(nbytes)
    ora = O.OracleSketcher(O.MASH, n, k, 0)
    ora.process_packed(host, 0)
    assert_same(sk, ora, "resident k=%d" % k)
    # reset + re-run gives the same answer (fh_reset clears exactly what was touched)
    kc1 = sk.to_arrays()
    sk.reset()
    sk.push_device(dr.ptr, nbytes)
    kc2 = sk.to_arrays()
    assert all(np.array_equal(a, b) for a, b in zip(kc1, kc2))


@pytest.mark.parametrize("n,inflight", [(3000, 65536), (1000, 16384), (20000, 262144)])
def test_capacity_stop_and_relaunch(n, inflight, monkeypatch):
    """a launch whose waves stop because the table nears its guarded size is pruned and relaunched from the
    tile queue: nothing lost, nothing counted twice"""
    # (the speculative threshold falls short on these reads; re-reading the block for EVERYTHING above it, as until round 4, is
    # what fills the table here -- the rescaled re-read of round 4 does not get that far)
    F.debug_set(no_spec_rescale="1")
    gl, nr, rl, seed = 500000, 200000, 150, 21
    dg = F.DeviceBuffer(gl)
    nbytes = nr * (rl + 1)
    dr = F.DeviceBuffer(nbytes + 64)
    S.synth_genome_device(dg, gl, seed)
    S.synth_reads_device(dr, dg, gl, 0, nr, rl, seed, 10000, 500)
    sk = F.SketchParams.mash(n, n, True, 21, 0).create_sketcher(max_launch=inflight)
    sk.push_device(dr.ptr, nbytes)
    ora = O.OracleSketcher(O.MASH, n, 21, 0)
    ora.process_packed(dr.download(nbytes), 0)
    assert_same(sk, ora, "stop/relaunch n=%d" % n)
    c = sk.debug_counters()
    assert c["relaunches"] >= 1, c


def test_speculative_threshold_and_its_second_pass():
    """first block of a fresh sketcher: threshold guessed from the block length; low-complexity input (fewer than
    `size` distinct k-mers below the guess) must trigger the second pass and still be bit-exact"""
    rng = np.random.default_rng(17)
    # (a) diverse input: the guess holds
    g = S.synth_genome_host(400000, 4)
    reads = S.synth_reads_host(g, 0, 20000, 150, 4, 10000, 500)
    sk = F.SketchParams.mash(1000, 1000, True, 21, 0).create_sketcher()
    sk.push_block(reads)
    ora = O.OracleSketcher(O.MASH, 1000, 21, 0)
    ora.process_packed(reads, 0)
    assert_same(sk, ora, "spec ok")
    c = sk.debug_counters()
    assert c["spec"] == 1 and c["spec_second_pass"] == 0, c
    # (b) 700 distinct k-mers repeated many times: the guess captures far fewer than 1000
    unit = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=720))
    block = (unit + b"\x00") * 400
    for kind, size in [("mash", 1000), ("scaled", 1000), ("mash", 20000)]:
        params = (F.SketchParams.mash(size, size, True, 21, 0) if kind == "mash" else F.SketchParams.scaled(size, 21, 1e-6, 0))
        sk = params.create_sketcher()
        sk.push_block(block)
        ora = O.OracleSketcher(O.MASH if kind == "mash" else O.SCALED, size, 21, 0, 1e-6)
        ora.process_packed(block, 0)
        assert_same(sk, ora, "spec second pass %s %d" % (kind, size))
        c = sk.debug_counters()
        assert c["spec"] == 1 and c["spec_second_pass"] == 1, c
        # a second block after that goes through the ordinary path
        sk2 = params.create_sketcher()
        sk2.push_block(block)
        sk2.push_block(reads)
        ora.process_packed(reads, 0)
        assert_same(sk2, ora, "spec second pass then more data")


def test_speculation_on_the_prefix_of_a_large_first_block():
    """a first block above 64 M positions speculates on its first 32 M positions only; if that prefix is
    low-complexity the second pass covers the prefix alone and the rest of the block continues normally"""
    k, n = 21, 1000
    rng = np.random.default_rng(23)
    g = S.synth_genome_host(12_000_000, 9)  # low coverage: most k-mers of the prefix are distinct
    reads = S.synth_reads_host(g, 0, 300000, 150, 9, 10000, 500)  # 45 MB of diverse reads
    # (a) diverse from the start: the guess holds, no second pass
    big = np.concatenate([reads, S.synth_reads_host(g, 300000, 250000, 150, 9, 10000, 500)])
    assert len(big) > (64 << 20) + 1000
    buf = F.DeviceBuffer(len(big) + 64)
    buf.upload(big)
    sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher()
    sk.push_device(buf.ptr, len(big))
    ora = O.OracleSketcher(O.MASH, n, k, 0)
    ora.process_packed(big, 0)
    assert_same(sk, ora, "prefix speculation ok")
    c = sk.debug_counters()
    assert c["spec"] == 1 and c["spec_second_pass"] == 0, c
    buf.free()
    # (b) 40 MB of a 720-base unit (700 distinct k-mers) in front of the diverse reads: the prefix holds fewer than
    # n distinct hashes below the guess -> second pass over the prefix only, then the ordinary path for the rest
    unit = np.frombuffer(bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=720)) + b"\x00", np.uint8)
    low = np.tile(unit, (40 << 20) // len(unit))
    mixed = np.concatenate([low, reads])
    assert len(mixed) > (64 << 20) + 1000 and len(low) > (32 << 20)
    for kind, size in [("mash", n), ("scaled", n)]:
        params = F.SketchParams.mash(size, size, True, k, 0) if kind == "mash" else F.SketchParams.scaled(size, k, 1e-6, 0)
        buf = F.DeviceBuffer(len(mixed) + 64)
        buf.upload(mixed)
        sk = params.create_sketcher()
        sk.push_device(buf.ptr, len(mixed))
        ora = O.OracleSketcher(O.MASH if kind == "mash" else O.SCALED, size, k, 0, 1e-6)
        ora.process_packed(mixed, 0)
        assert_same(sk, ora, "prefix speculation, second pass (%s)" % kind)
        c = sk.debug_counters()
        assert c["spec"] == 1 and c["spec_second_pass"] == 1, c
        buf.free()


def test_low_diversity_stream_does_not_crawl_through_tiny_ranges():
    """small k on a long stream: once every distinct k-mer is in the table each admitted occurrence is a duplicate, so
    the live set stops growing; ranges must be sized from the observed novelty, not from the admit rate alone
    (k = 8 used to take ~280 closed-loop launches per 32 M positions)"""
    n = 1000
    g = S.synth_genome_host(12_000_000, 31)
    reads = S.synth_reads_host(g, 0, 470000, 150, 31, 10000, 500)  # ~70 M positions
    assert len(reads) > (64 << 20)
    buf = F.DeviceBuffer(len(reads) + 64)
    buf.upload(reads)
    for k in (7, 8, 11):
        sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher()
        sk.push_device(buf.ptr, len(reads))
        ora = O.OracleSketcher(O.MASH, n, k, 0)
        ora.process_packed(reads, 0)
        assert_same(sk, ora, "low diversity k=%d" % k)
        c = sk.debug_counters()
        assert c["launches"] <= 40, (k, c)
    buf.free()


def test_copy_out_forms_agree():
    """fh_copy_out (separate arrays) and fh_copy_out_records (KmerCount-shaped records) return the same sketch,
    inline (small) and threaded (>= 128 k records), before and after a merge"""
    import ctypes as C
    g = S.synth_genome_host(3_000_000, 77)
    for n in (500, 300_000):
        sk = F.SketchParams.mash(n, n, True, 21, 0).create_sketcher()
        sk.push_block(g)
        other = F.SketchParams.mash(n, n, True, 21, 0).create_sketcher()
        other.push_block(S.synth_genome_host(1_000_000, 78))
        other.finish()
        for merged in (False, True):
            if merged:
                sk.merge(other)
            kc, km, ps = sk.to_arrays()
            m = len(kc)
            assert m == n
            hs, cs, es = np.zeros(m, np.uint64), np.zeros(m, np.uint32), np.zeros(m, np.uint32)
            km2, ps2 = np.zeros((m, 21), np.uint8), np.zeros(m, np.uint64)
            P = lambda a: a.ctypes.data_as(C.c_void_p)
            S.check(sk._L.fh_copy_out(sk._h, P(hs), P(cs), P(es), P(km2), P(ps2)))
            assert (kc["hash"] == hs).all() and (kc["count"] == cs).all() and (kc["extra_count"] == es).all()
            assert (km == km2).all() and (ps == ps2).all()
            assert (np.diff(hs.astype(np.float64)) > 0).all()


def test_select_prune_equals_sort_prune():
    """between launches large live sets are pruned by a radix select; option no_select makes every prune the full
    sort that fh_finish uses.  Both must give the same sketch (run in a subprocess: the switch is read once)."""
    import subprocess
    import sys
    code = (
        "import numpy as np, finch_rs_amd as F\n"
        "from finch_rs_amd import sketch_schemes as S\n"
        "g = S.synth_genome_host(1500000, 31)\n"
        "r = S.synth_reads_host(g, 0, 150000, 150, 31, 10000, 500)\n"
        "out = []\n"
        "for p in (F.SketchParams.mash(60000, 60000, True, 25, 3), F.SketchParams.scaled(30000, 19, 0.002, 0)):\n"
        "    sk = p.create_sketcher(); sk.push_block(r); kc, km, pos = sk.to_arrays()\n"
        "    out.append((kc['hash'].sum(dtype=np.uint64), int(kc['count'].sum()), int(kc['extra_count'].sum()), len(kc),\n"
        "                int(km.astype(np.uint64).sum()), sk.debug_counters()['big_prunes']))\n"
        "print(repr(out))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for extra in ({}, {"no_select": "1"}):
        env = F.debug_env(**extra)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(eval(r.stdout.strip().splitlines()[-1], {"np": np}))
    assert res[0] == res[1], res
    assert all(x[5] >= 2 for x in res[0]), res[0]  # the in-stream big prune really ran


def test_sharded_merge_equals_whole():
    """global sketch == merge of read-block shard sketches"""
    gl, nr, rl, seed = 300000, 120000, 150, 7
    g = S.synth_genome_host(gl, seed)
    reads = S.synth_reads_host(g, 0, nr, rl, seed, 10000, 500)
    rec = rl + 1
    whole = F.SketchParams.mash(500, 500, True, 21, 0).create_sketcher()
    whole.push_block(reads)
    parts = []
    cuts = [0, 30000, 30001, 90000, nr]
    for a, b in zip(cuts[:-1], cuts[1:]):
        p = F.SketchParams.mash(500, 500, True, 21, 0).create_sketcher()
        p.set_stream_offset(a * rec)
        p.push_block(reads[a * rec:b * rec])
        p.finish()
        parts.append(p)
    merged = parts[0]
    for p in parts[1:]:
        merged.merge(p)
    a, b = whole.to_arrays(), merged.to_arrays()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert whole.finish()[1] == merged.finish()[1]
    ora = O.OracleSketcher(O.MASH, 500, 21, 0)
    ora.process_packed(reads, 0)
    assert_same(merged, ora)


def test_hash_collisions_keep_first_kmer():
    """64-bit collisions between distinct k-mers, forced with the hash_mask test hook:
    counts are summed and the k-mer bytes of the FIRST occurrence are kept (mash.rs:45-56)."""
    rng = np.random.default_rng(3)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=40000)
    reads = random_reads(rng, 500, 50, 150, genome=genome)
    block = b"".join(r + b"\x00" for r in reads)
    n_coll_members = 0
    for mask in (0xFFFFF, 0x3FFFF):
        sk = F.SketchParams.mash(300, 300, True, 21, 0).create_sketcher(hash_mask=mask)
        sk.push_block(block)
        ora = O.OracleSketcher(O.MASH, 300, 21, 0)
        ora.set_hash_mask(mask)
        ora.process_packed(block, 0)
        assert_same(sk, ora, "masked %x" % mask)
        # make sure the case is actually exercised: some retained hash stands for >1 distinct k-mer
        full = O.OracleSketcher(O.MASH, 10**6, 21, 0)
        full.process_packed(block, 0)
        fk, _ = full.to_vec()
        masked = fk["hash"] & np.uint64(mask)
        vals, cnt = np.unique(masked, return_counts=True)
        dup = set(vals[cnt > 1].tolist())
        n_coll_members += sum(1 for h in ora.to_vec()[0]["hash"].tolist() if h in dup)
    assert n_coll_members > 0


@pytest.mark.parametrize("k", [33, 48, 63, 64])
def test_two_word_kmers_full_feature_parity(k):
    """k = 33..64 (finch's kmer_length is a u8; mod.rs:54-71): the two-word kernel through everything the one-word path
    is tested with -- resident streams with a capacity-limited table, large sketches (device-wide selection), scaled
    sketches incl. table growth, sharded merge, forced 64-bit hash collisions (k-mer bytes of the first occurrence)."""
    g = S.synth_genome_host(400000, 50 + k)
    reads = S.synth_reads_host(g, 0, 20000, 150, 50 + k, 10000, 500)
    for n, inflight in ((1000, 0), (500, 8192), (20000, 0)):
        sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher(max_launch=inflight)
        sk.push_block(reads)
        ora = O.OracleSketcher(O.MASH, n, k, 0)
        ora.process_packed(reads, 0)
        assert_same(sk, ora, "k=%d n=%d inflight=%d" % (k, n, inflight))
    # resident block, seed != 0
    db = F.DeviceBuffer(len(reads) + 64)
    db.upload(reads)
    sk = F.SketchParams.mash(300, 300, True, k, 42).create_sketcher()
    sk.push_device(db.ptr, len(reads))
    ora = O.OracleSketcher(O.MASH, 300, k, 42)
    ora.process_packed(reads, 0)
    assert_same(sk, ora, "resident k=%d" % k)
    # scaled, incl. an unbounded one that outgrows its table
    for size, scale in ((100, 0.001), (0, 0.2)):
        sk = F.SketchParams.scaled(size, k, scale, 0).create_sketcher()
        sk.push_block(reads)
        ora = O.OracleSketcher(O.SCALED, size, k, 0, scale)
        ora.process_packed(reads, 0)
        assert_same(sk, ora, "scaled k=%d %r" % (k, scale))
    # two read blocks on two handles, merged on the host
    half = 10000 * 151
    a = F.SketchParams.mash(700, 700, True, k, 0).create_sketcher()
    b = F.SketchParams.mash(700, 700, True, k, 0).create_sketcher()
    a.push_block(reads[:half])
    b.set_stream_offset(half)
    b.push_block(reads[half:])
    a.finish(); b.finish()
    a.merge(b)
    ora = O.OracleSketcher(O.MASH, 700, k, 0)
    ora.process_packed(reads, 0)
    assert_same(a, ora, "merged k=%d" % k)
    # forced collisions: distinct k-mers sharing a (masked) hash -- and, for good measure, k-mers that share their last
    # 32 bases (a repeat planted in the reads), which is what the two-word compare has to tell apart
    rep = bytes(S.synth_genome_host(100, 7))
    block = reads[:3000 * 151].tobytes() + b"".join(bytes(g[i * 100:i * 100 + 40]) + rep[:70] + b"\x00" for i in range(400))
    for mask in (0xFFFFF, 0x3FFF):
        sk = F.SketchParams.mash(300, 300, True, k, 0).create_sketcher(hash_mask=mask)
        sk.push_block(block)
        ora = O.OracleSketcher(O.MASH, 300, k, 0)
        ora.set_hash_mask(mask)
        ora.process_packed(block, 0)
        assert_same(sk, ora, "masked %x k=%d" % (mask, k))


def test_the_one_64mer_whose_low_word_looks_unclaimed():
    """k = 64: the table marks an unclaimed k-mer word with all ones, which is also the low word of a 64-mer that ends in 32 T
    (only A^32 T^32 is canonical with it).  Its occurrences, and a second k-mer forced onto the same hash, must come out
    with the bytes of the first occurrence (mash.rs:52-56), whichever order they arrive in."""
    special = b"A" * 32 + b"T" * 32
    other = bytes(S.synth_genome_host(64, 99))
    for recs in ([special] * 5 + [other] * 3, [other] * 2 + [special] * 4, [special, other, special]):
        block = b"".join(r + b"\x00" for r in recs)
        for mask in (0, 0x1):  # mask 1: nearly everything collides
            sk = F.SketchParams.mash(10, 10, True, 64, 0).create_sketcher(hash_mask=mask)
            sk.push_block(block)
            ora = O.OracleSketcher(O.MASH, 10, 64, 0)
            if mask:
                ora.set_hash_mask(mask)
            ora.process_packed(block, 0)
            assert_same(sk, ora, "special 64-mer, mask %x" % mask)


def test_large_sketch_takes_its_threshold_from_a_sample():
    """kmers_to_sketch in the tens of thousands on a large first block: the threshold is estimated from a sparse sample of
    the block (k_sample_hashes) and the block sketched in one launch; an estimate that is too tight is repaired by a
    second pass for the hashes above it; one that is far too loose only costs time.  Bit-exact in every case, also for
    streams the estimator cannot read (few distinct k-mers).  The knobs are read once per process: child processes."""
    code = r'''
import os, numpy as np
import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O
g = S.synth_genome_host(3_000_000, 77)
reads = S.synth_reads_host(g, 0, 1_000_000, 150, 77, 10000, 500)   # 150 Mbase, 50 x coverage, 1 % errors
db = F.DeviceBuffer(len(reads) + 64); db.upload(reads)
expect = os.environ["EXPECT"]
for k, n in ((21, 20000), (31, 100000)) if expect == "hit" else ((21, 20000),):
    sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher()
    sk.push_device(db.ptr, len(reads))
    kc, km, _ = sk.to_arrays()
    ora = O.OracleSketcher(O.MASH, n, k, 0); ora.process_packed(reads, 0)
    okc, okm = ora.to_vec()
    assert np.array_equal(kc, okc) and np.array_equal(km, okm) and sk.finish()[1] == ora.total_bases_and_kmers()[1], (k, n)
    c = sk.debug_counters()
    if expect == "hit":  # (a guess that lands just below the final threshold is repaired: rare, not wrong)
        assert c["spec"] == 1 and c["spec_second_pass"] in (0, 1), c
    elif expect == "repair":
        assert c["spec"] == 1 and c["spec_second_pass"] == 1, c
# a stream of few distinct k-mers (one read over and over): nothing to estimate from, same result
rep = np.tile(reads[:151 * 40], 25000)
db2 = F.DeviceBuffer(len(rep) + 64); db2.upload(rep)
sk = F.SketchParams.mash(20000, 20000, True, 21, 0).create_sketcher()
sk.push_device(db2.ptr, len(rep))
ora = O.OracleSketcher(O.MASH, 20000, 21, 0); ora.process_packed(rep, 0)
kc, km, _ = sk.to_arrays(); okc, okm = ora.to_vec()
assert np.array_equal(kc, okc) and np.array_equal(km, okm)
print("child ok")
'''
    import subprocess, sys
    for env in ({"EXPECT": "hit"}, {"EXPECT": "repair", "sample_scale": "0.02"}, {"EXPECT": "loose", "sample_scale": "30"},
                {"EXPECT": "off", "no_sample": "1"}):
        e = F.debug_env(sample_min_pos="1000000", **{k: v for k, v in env.items() if k.islower()})
        e["EXPECT"] = env["EXPECT"]
        r = subprocess.run([sys.executable, "-c", code], env=e, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0 and "child ok" in r.stdout, (env, r.stdout[-3000:])


def test_handle_cache_returns_a_clean_sketcher():
    """fh_free parks the reset handle, fh_new with the same parameters takes it over: the second owner must see a
    fresh sketcher (empty, counters at zero, same results as a brand-new one); different parameters get their own
    handle; fh_release_cached empties the cache"""
    from finch_rs_amd import _lib
    g = S.synth_genome_host(200000, 21)
    reads = S.synth_reads_host(g, 0, 8000, 150, 21, 10000, 500)
    other = S.synth_reads_host(g, 8000, 3000, 150, 21, 10000, 500)
    p = F.SketchParams.mash(500, 500, True, 21, 0)
    a = p.create_sketcher()
    a.push_block(other)
    a.finish()
    assert a.debug_counters()["launches"] > 0
    a.close()                      # parked
    b = p.create_sketcher()        # the same handle again
    c = b.debug_counters()
    assert c["launches"] == 0 and c["spec"] == 0, c
    assert b.finish() == (0, 0)    # empty sketch, state as after fh_new
    b.reset()
    b.push_block(reads)
    ora = O.OracleSketcher(O.MASH, 500, 21, 0)
    ora.process_packed(reads, 0)
    assert_same(b, ora, "recycled handle")
    q = F.SketchParams.mash(500, 500, True, 31, 7).create_sketcher()   # other parameters: not the parked one
    q.push_block(reads)
    ora2 = O.OracleSketcher(O.MASH, 500, 31, 7)
    ora2.process_packed(reads, 0)
    assert_same(q, ora2, "different parameters")
    b.close()
    q.close()
    _lib.load().fh_release_cached()
    d = p.create_sketcher()        # a new allocation after the cache was emptied
    d.push_block(reads)
    assert_same(d, ora, "after fh_release_cached")
    d.close()
