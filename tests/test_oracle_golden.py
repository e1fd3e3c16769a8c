"""Pin the CPU oracle (oracle/) against every known-answer vector the reference's own tests hold
for the sketching hot path.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(scope="module")
def vec(golden_dir):
    with open(os.path.join(golden_dir, "reference_vectors.json")) as f:
        return json.load(f)


def _kmers(km):
    return [bytes(r).decode() for r in km]


def test_murmur3_published_vectors():
    # MurmurHash3_x64_128 known answers (SMHasher reference implementation), 32-bit seeds
    assert O.murmur3_x64_128(b"", 0) == (0, 0)
    h1, h2 = O.murmur3_x64_128(b"hello", 0)
    assert (h1, h2) == (0xCBD8A7B341BD9B02, 0x5B1E906A48AE1D19)
    h1, h2 = O.murmur3_x64_128(b"The quick brown fox jumps over the lazy dog", 0)
    assert (h1, h2) == (0xE34BBC7BBC071B6C, 0x7A433CA9C49A9347)


@pytest.mark.parametrize("kind", [O.MASH, O.SCALED])
def test_minhashkmers_seed42(vec, kind):
    # mash.rs:115-134, scaled.rs:118-138 (scale 1.) and 140-161 (scale .001, size 3)
    v = vec["mash_rs_116_134"]
    for scale in ([1.0] if kind == O.MASH else [1.0, 0.001]):
        s = O.OracleSketcher(kind, v["size"], v["k"], v["seed"], scale)
        for kmer, extra in v["pushes"]:
            s.push(kmer.encode(), extra)
        kc, km = s.to_vec()
        assert _kmers(km) == v["expect_order"]
        assert [[int(c), int(e)] for c, e in zip(kc["count"], kc["extra_count"])] == v["expect_counts"]
        assert np.all(np.diff(kc["hash"].astype(object)) > 0)


def test_longer_sequence_hashes(vec):
    # mash.rs:136-154: the 11 hashes are those of the *canonical* k-mers
    v = vec["mash_rs_141_153"]
    s = O.OracleSketcher(O.MASH, 100, v["k"], v["seed"])
    s.process(v["sequence"].encode())
    kc, km = s.to_vec()
    assert [str(int(h)) for h in kc["hash"]] == v["hashes"]
    assert s.total_bases_and_kmers() == (31, 11)


def test_scaled_eviction(vec):
    v = vec["scaled_rs_163_176"]
    s = O.OracleSketcher(O.SCALED, v["size"], v["k"], v["seed"], v["scale"])
    for kmer, extra in v["pushes"]:
        s.push(kmer.encode(), extra)
    kc, km = s.to_vec()
    assert len(kc) == v["expect_len"]
    assert v["expect_absent"] not in _kmers(km)
    assert O.hash_f(b"AAAA", 42) > s.max_hash


@pytest.mark.parametrize("size", [0])
def test_pure_scaled_empty(size):
    # scaled.rs:178-200
    s = O.OracleSketcher(O.SCALED, size, 2, 42, 0.001)
    for kmer, extra in [(b"ca", 0), (b"cc", 1), (b"ac", 0), (b"ac", 1)]:
        s.push(kmer, extra)
    assert len(s.to_vec()[0]) == 0


def test_pure_scaled_property():
    # scaled.rs:202-213 (proptest pure_scaled_check), seeded here
    rng = np.random.default_rng(1234)
    for _ in range(20):
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(500, 900))))
        s = O.OracleSketcher(O.SCALED, 0, 2, 42, 1.0 / 100.0)
        for i in range(len(seq) - 3):
            s.push(seq[i:i + 4], 0)
        kc, _ = s.to_vec()
        assert np.all(kc["hash"] <= np.uint64((2**64 - 1) // 100))


@pytest.mark.parametrize("kind", [O.MASH, O.SCALED])
def test_cli_golden_kmers(vec, golden_dir, kind):
    # cli/tests/test_cli.rs:80-149 : first 10 k-mers in ascending-hash order, mash and scaled identical
    v = vec["test_cli_rs_99_143"]
    data = open(os.path.join(golden_dir, v["file"]), "rb").read()
    s = O.OracleSketcher(kind, v["n"], v["k"], v["seed"], v["scale"])
    fmt = s.sketch_stream(data)
    assert fmt == 1  # FASTA
    kc, km = s.to_vec()
    assert _kmers(km)[:10] == v["kmers"]
    # companions derived in SURVEY.md 8c (O3): hashes / counts / extra / numValidKmers
    assert [int(h) for h in kc["hash"][:10]] == [
        933085113509804, 8582128962097342, 12581283643378369, 13388215406653903, 59671498055219043,
        85163822212241463, 196329111101504065, 240583695071237384, 241465901919730030, 256930375650047524]
    assert [int(c) for c in kc["count"][:10]] == [1, 1, 1, 1, 1, 1, 2, 1, 1, 2]
    assert [int(c) for c in kc["extra_count"][:10]] == [0, 0, 1, 1, 1, 1, 2, 1, 1, 0]
    assert s.total_bases_and_kmers()[1] == 339


def test_normalize_semantics():
    assert O.normalize(b"ACGTacgtuUnN.-~ \t\r\nRYxz*") == b"ACGTACGTTTNN---NNNNN"
    assert O.reverse_complement(b"ACGTN-") == b"-NACGT"


def test_whitespace_is_skipped_not_a_breaker():
    a = O.OracleSketcher(O.MASH, 1000, 5, 0)
    a.process(b"ACGTT\nGCAAT\r\nCCGA")
    b = O.OracleSketcher(O.MASH, 1000, 5, 0)
    b.process(b"ACGTTGCAATCCGA")
    ka, kb = a.to_vec(), b.to_vec()
    assert np.array_equal(ka[0], kb[0]) and np.array_equal(ka[1], kb[1])
    assert a.total_bases_and_kmers() == (17, 10) and b.total_bases_and_kmers() == (14, 10)


def test_order_independence_and_merge_property():
    # the result is a function of the multiset of k-mers
    rng = np.random.default_rng(7)
    reads = [bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), p=[.24, .24, .24, .24, .04], size=80))
             for _ in range(300)]
    reads = reads + reads[:100]
    for kind, size in [(O.MASH, 50), (O.SCALED, 50), (O.SCALED, 0)]:
        s1 = O.OracleSketcher(kind, size, 11, 0, 0.05)
        for r in reads:
            s1.process(r)
        s2 = O.OracleSketcher(kind, size, 11, 0, 0.05)
        for i in rng.permutation(len(reads)):
            s2.process(reads[i])
        a, b = s1.to_vec(), s2.to_vec()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_config0_fasta_fingerprint_is_what_the_oracle_gives(golden_dir):
    """BASELINE configs[0]: the committed golden c1_fasta_k21_n1000 is the oracle's sketch_stream of genome G as 70-column FASTA
    (the generator script re-run here: seconds on one core)"""
    import sys
    sys.path.insert(0, golden_dir)
    import make_config_fingerprints as M
    text = M.fasta_70(M.genome_numpy(M.GL, M.SEED))
    o = O.OracleSketcher(O.MASH, 1000, 21, 0)
    assert o.sketch_stream(text) == 1
    kc, km = o.to_vec()
    tb, tk = o.total_bases_and_kmers()
    golden = json.load(open(os.path.join(golden_dir, "config_fingerprints.json")))["c1_fasta_k21_n1000"]
    fp = M.fingerprint(kc, km, tk)
    assert all(fp[key] == golden[key] for key in fp) and tb == golden["seq_length"] and len(text) == golden["file_bytes"]
    # the numpy restatement of the genome generator agrees with the product's (which the GPU tests use for the big configs)
    from finch_rs_amd import sketch_schemes as S
    assert np.array_equal(M.genome_numpy(200_000, M.SEED), S.synth_genome_host(200_000, M.SEED))


def test_read_generator_restated_in_numpy_equals_the_products(golden_dir):
    """the golden fingerprints of configs[1..3] come from reads made by the product's host generator; here that generator is
    held against a numpy restatement of SURVEY 8d M4's rules written for the golden script (tests/golden/make_config_fingerprints.py)
    -- read blocks from the start, the middle and the end of configs[3]'s 333 M reads, and other lengths / rates"""
    import sys
    sys.path.insert(0, golden_dir)
    import make_config_fingerprints as M
    from finch_rs_amd import sketch_schemes as S
    g = M.genome_numpy(M.GL, M.SEED)
    for first, n, rl, sub, nn in ((0, 3000, 150, 10_000, 500), (166_000_000, 2000, 150, 10_000, 500), (333_333_000, 334, 150, 10_000, 500),
                                  (12345, 500, 61, 300_000, 20_000), (7, 200, 250, 0, 0)):
        a = M.reads_numpy(g, first, n, rl, M.SEED, sub, nn)
        b = S.synth_reads_host(g, first, n, rl, M.SEED, sub, nn)
        assert np.array_equal(a, b), (first, n, rl)
