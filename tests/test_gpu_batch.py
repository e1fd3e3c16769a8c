"""Many sketches per launch (fh_batch_*, fh_k2b.hip): every file the batch path TAKES carries the oracle's sketch bit for bit
-- hashes, counts, extra_counts, k-mer bytes, total k-mers (mash.rs:34-63, 86-102) -- and every file it does not take is one
the contract names (too few distinct k-mers below the threshold, ...), which the caller then sketches through a HipSketcher.
Through the C ABI; needs a real MI355X (`-m gpu`)."""
import numpy as np
import pytest

import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def oracle_sketch(block, n, k, seed):
    ora = O.OracleSketcher(O.MASH, n, k, seed)
    ora.process_packed(np.frombuffer(block, dtype=np.uint8) if not isinstance(block, np.ndarray) else block, 0)
    okc, okm = ora.to_vec()
    return okc, okm, ora.total_bases_and_kmers()[1]


def same(res, block, n, k, seed, ctx=""):
    kc, km, _, tk = res
    okc, okm, otk = oracle_sketch(block, n, k, seed)
    assert len(kc) == len(okc), (ctx, len(kc), len(okc))
    assert np.array_equal(kc["hash"], okc["hash"]), ctx
    assert np.array_equal(kc["count"], okc["count"]), ctx
    assert np.array_equal(kc["extra_count"], okc["extra_count"]), ctx
    assert np.array_equal(km, okm), ctx
    assert tk == otk, (ctx, tk, otk)


def genome_block(rng, length, n_records=1, p_n=0.0005, p_lower=0.0):
    """a packed stream: n_records records of random bases, one breaker byte behind each"""
    parts = []
    per = max(1, length // n_records)
    for _ in range(n_records):
        r = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=per)
        m = rng.random(per)
        r[m < p_n] = ord("N")
        if p_lower:
            low = m > 1 - p_lower
            r[low] = r[low] | 0x20
        parts.append(r)
        parts.append(np.zeros(1, np.uint8))
    return np.concatenate(parts)


@pytest.mark.parametrize("k,n,seed", [(21, 1000, 0), (21, 1000, 42), (31, 1000, 0), (16, 500, 0), (32, 3000, 7), (11, 100, 0), (24, 2000, 0)])
def test_batch_of_genomes_matches_oracle(k, n, seed):
    rng = np.random.default_rng(k * 1000 + n + seed)
    lens = [int(x) for x in rng.integers(150_000, 900_000, size=11)] + [2_000_000, 65_536, 2048 * 7, 2048 * 7 + 1]
    blocks = [genome_block(rng, L, n_records=int(rng.integers(1, 6)), p_lower=0.01) for L in lens]
    b = F.BatchSketcher(n, k, seed, max_files=8, stage_bytes=8 << 20)
    res = b.sketch_many(blocks)
    assert len(res) == len(blocks)
    taken = 0
    for i, (r, blk) in enumerate(zip(res, blocks)):
        if r is None:
            continue
        taken += 1
        same(r, blk, n, k, seed, "file %d (%d bytes)" % (i, len(blk)))
    # random genomes far longer than the sketch: the guess of ~4 n hashes below the threshold holds for every one
    assert taken == len(blocks), b.counters()
    b.close()


def test_small_empty_and_degenerate_files():
    rng = np.random.default_rng(5)
    k, n = 21, 1000
    blocks = [
        np.zeros(0, np.uint8),                                   # an empty file
        np.frombuffer(b"ACGT\0", dtype=np.uint8),                # shorter than k
        genome_block(rng, 900, 1),                               # fewer k-mers than n: everything is admitted, all of it kept
        genome_block(rng, 3999, 3),
        genome_block(rng, 4001, 1),                              # just above 4 n positions: a threshold, ~4000 expected below it
        genome_block(rng, 30_000, 1),
        np.frombuffer(b"N" * 5000 + b"\0", dtype=np.uint8),      # no valid window at all
        np.tile(np.frombuffer(b"ACGTTGCATGCATGACCA", dtype=np.uint8), 20000),  # 360 kb of an 18-base repeat: 18 distinct k-mers
        np.tile(genome_block(rng, 5000, 1), 100),                # 500 kb holding ~5000 distinct k-mers, a hundred times each
        genome_block(rng, 300_000, 1),
    ]
    b = F.BatchSketcher(n, k, 0, max_files=16, stage_bytes=4 << 20)
    res = b.sketch_many(blocks)
    for i, (r, blk) in enumerate(zip(res, blocks)):
        if r is not None:
            same(r, blk, n, k, 0, "file %d" % i)
    # what MUST be taken: the files whose threshold admits everything, and the plain genomes
    for i in (0, 1, 2, 3, 5, 9):
        assert res[i] is not None, i
    # what must NOT be: fewer than n distinct k-mers below the threshold the file was sketched at (file 6: none at all behind
    # a threshold -- 5001 positions are more than the 4 n that go without one; file 7: 18 k-mers in all)
    assert res[6] is None and res[7] is None
    # ... and those go through a HipSketcher, which is exact for anything
    for i, r in enumerate(res):
        if r is None:
            sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher()
            sk.push_block(blocks[i])
            kc, km, _ = sk.to_arrays()
            okc, okm, otk = oracle_sketch(blocks[i], n, k, 0)
            assert np.array_equal(kc, okc) and np.array_equal(km, okm) and sk.finish()[1] == otk
    c = b.counters()
    assert c["taken"] + c["not_taken"] == len(blocks)
    b.close()


def test_two_slots_alternate_and_partitions_come_back_clean():
    """batch after batch through both slots: a partition that held file A's hashes must hold nothing of them when file B
    comes to it (the epilogue leaves every partition reset), including after files that were not taken"""
    rng = np.random.default_rng(11)
    k, n = 21, 1000
    b = F.BatchSketcher(n, k, 0, max_files=4, stage_bytes=2 << 20)
    rounds = []
    for r in range(6):
        blocks = [genome_block(rng, int(rng.integers(50_000, 400_000)), int(rng.integers(1, 4))) for _ in range(4)]
        if r % 2:
            blocks[1] = np.tile(np.frombuffer(b"ACGTTGCATGCATGACCATT", dtype=np.uint8), 5000)  # not taken
        rounds.append(blocks)
    pending = None
    results = []
    for r, blocks in enumerate(rounds):
        slot = r & 1
        buf = b.stage(slot)
        offs, lens, pos = [], [], 0
        for blk in blocks:
            buf[pos:pos + len(blk)] = blk
            offs.append(pos)
            lens.append(len(blk))
            pos = (pos + len(blk) + 15) & ~15
        b.submit(slot, offs, lens)
        if pending is not None:
            ps, pn = pending
            st = b.wait(ps, pn)
            results.append([b.result(ps, j) if st[j] == 0 else None for j in range(pn)])
        pending = (slot, len(blocks))
    ps, pn = pending
    st = b.wait(ps, pn)
    results.append([b.result(ps, j) if st[j] == 0 else None for j in range(pn)])
    for r, (blocks, res) in enumerate(zip(rounds, results)):
        for j, (blk, x) in enumerate(zip(blocks, res)):
            if r % 2 and j == 1:
                assert x is None
            else:
                assert x is not None, (r, j)
                same(x, blk, n, k, 0, "round %d file %d" % (r, j))
    b.close()


def test_argument_errors():
    with pytest.raises(F.FinchHipError):
        F.BatchSketcher(5000, 21)  # more hashes than the in-LDS selection serves
    with pytest.raises(F.FinchHipError):
        F.BatchSketcher(1000, 33)  # two-word k-mers go through HipSketcher
    b = F.BatchSketcher(1000, 21, max_files=2, stage_bytes=1 << 20)
    with pytest.raises(F.FinchHipError):
        b.submit(0, [8], [100])  # not 16-byte aligned
    with pytest.raises(F.FinchHipError):
        b.submit(0, [0], [(1 << 20) + 4096 + 1])  # beyond the staging buffer
    with pytest.raises(F.FinchHipError):
        b.submit(0, [0, 16, 32], [1, 1, 1])  # more files than the handle serves
    with pytest.raises(F.FinchHipError):
        b.wait(0, 1)  # nothing in flight
    b.submit(0, [], [])
    assert len(b.wait(0, 0)) == 0
    b.close()


# --- the host layer on top: finch_sketch_files stages plain FASTA files in groups (fh_host.cpp) ---
import gzip  # noqa: E402

from finch_rs_amd import host as H  # noqa: E402
from finch_rs_amd.sketch_schemes import FinchError, SketchParams  # noqa: E402


def _fasta(seq: bytes, name=b"g", width=70, eol=b"\n", last_eol=True):
    body = eol.join(seq[j:j + width] for j in range(0, len(seq), width))
    return b">" + name + eol + body + (eol if last_eol else b"")


def _same_as_oracle(sk, data, n, k, seed=0, keep=None):
    o = O.OracleSketcher(O.MASH, n, k, seed)
    o.sketch_stream(data)
    okc, okm = o.to_vec()
    if keep is not None:
        okc, okm = okc[:keep], okm[:keep]
    assert np.array_equal(sk.arrays[0], okc) and np.array_equal(sk.arrays[1], okm)
    assert (sk.seq_length, sk.num_valid_kmers) == o.total_bases_and_kmers()


def test_sketch_files_groups_match_the_oracle_and_the_one_by_one_path(tmp_path, monkeypatch):
    """a mixed batch: plain genomes (grouped), CRLF and unterminated files, several contigs, a repeat (not taken), a file
    with too few k-mers for the sketch only in no_strict mode, FASTQ and gzip'd FASTA (never grouped) -- one sketch per file,
    input order (lib.rs:29-49), each equal to the oracle's sketch_stream and to what option file_batch=0 gives"""
    rng = np.random.default_rng(21)
    datas = []
    for i in range(20):
        L = int(rng.integers(30_000, 600_000))
        datas.append(_fasta(bytes(S.synth_genome_host(L, 500 + i)), b"g%d len=%d" % (i, L)))
    datas.append(_fasta(bytes(S.synth_genome_host(90_000, 7)), eol=b"\r\n"))
    datas.append(_fasta(bytes(S.synth_genome_host(77_777, 8)), last_eol=False))
    datas.append(b"".join(_fasta(bytes(S.synth_genome_host(int(rng.integers(100, 40_000)), 900 + c)), b"contig%d" % c, width=60) for c in range(9)))
    datas.append(_fasta(b"ACGTTGCATGCATGACCATT" * 20_000))                        # 400 kb, 20 distinct k-mers: not taken by the batch path
    datas.append(_fasta(bytes(S.synth_genome_host(700, 9))))                       # 680 k-mers: fewer than the sketch holds
    reads = S.synth_reads_host(S.synth_genome_host(50_000, 3), 0, 3000, 100, 1, 5000, 500)
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(reads[i * 101:i * 101 + 100]), b"I" * 100) for i in range(3000))
    datas.append(fq)                                                               # FASTQ: filtering on by default, its own sketcher
    datas.append(gzip.compress(datas[0], 1))
    paths = []
    for i, d in enumerate(datas):
        p = tmp_path / ("f%02d" % i)
        p.write_bytes(d)
        paths.append(str(p))
    params = SketchParams.mash(1000, 1000, True, 21, 0)
    t0, n0 = H.debug_file_batch()
    res = H.sketch_files(paths, params, H.FilterParams(None), n_threads=3)
    t1, n1 = H.debug_file_batch()
    assert len(res) == len(paths)
    assert t1 - t0 >= 22 and n1 - n0 >= 1, (t1 - t0, n1 - n0)   # the genomes went many-per-launch, the repeat did not
    F.debug_set(file_batch="0")
    ref = H.sketch_files(paths, params, H.FilterParams(None), n_threads=3)
    assert H.debug_file_batch() == (t1, n1)
    for i, d in enumerate(datas):
        a, b = res.sketch(i), ref.sketch(i)
        assert a.name == b.name == paths[i]
        assert np.array_equal(a.arrays[0], b.arrays[0]) and np.array_equal(a.arrays[1], b.arrays[1]), i
        assert (a.seq_length, a.num_valid_kmers) == (b.seq_length, b.num_valid_kmers), i
        assert a.filter_params == b.filter_params and a.sketch_params == b.sketch_params, i
        if i < len(datas) - 2:
            _same_as_oracle(a, d, 1000, 21)
    # the groups with bytes on the link instead of the two-bit form (fh_batch_submit_packed is the default), and the two-bit
    # form written by the packer's portable code: the same sketches, the same files taken
    # ... and the files read in pieces of 4099 and 5000 bytes (header lines, line ends and CR LF pairs cut by piece ends)
    for opts in (dict(batch_two_bit="0"), dict(pack_scalar="1"), dict(pack_scalar="2"), dict(batch_read_piece="4099"), dict(batch_read_piece="5000")):
        F.debug_set(file_batch=None, **opts)
        t2, n2 = H.debug_file_batch()
        alt = H.sketch_files(paths, params, H.FilterParams(None), n_threads=3)
        t3, n3 = H.debug_file_batch()
        assert (t3 - t2, n3 - n2) == (t1 - t0, n1 - n0), opts
        for i in range(len(datas)):
            a, b = res.sketch(i), alt.sketch(i)
            assert np.array_equal(a.arrays[0], b.arrays[0]) and np.array_equal(a.arrays[1], b.arrays[1]), (opts, i)
            assert (a.seq_length, a.num_valid_kmers) == (b.seq_length, b.num_valid_kmers), (opts, i)
        F.debug_set(**{k: None for k in opts})
    # strict mode: the file with 680 k-mers is the reference's error (mod.rs:123-125), whichever path took it
    F.debug_set(file_batch=None)
    with pytest.raises(FinchError, match="had too few kmers \\(680\\) to sketch"):
        H.sketch_files(paths[:23] + [paths[24]], SketchParams.mash(1000, 1000, False, 21, 0), H.FilterParams(None), n_threads=2)


def test_sketch_files_groups_with_oversketch_seed_and_other_k(tmp_path):
    """final_size < kmers_to_sketch without filtering: the group's sketches are of final_size hashes (what the small sketcher
    of sketch_stream makes); a seed; k = 31 and k = 12"""
    rng = np.random.default_rng(22)
    datas = [_fasta(bytes(S.synth_genome_host(int(rng.integers(50_000, 300_000)), 40 + i))) for i in range(10)]
    paths = []
    for i, d in enumerate(datas):
        p = tmp_path / ("o%02d.fa" % i)
        p.write_bytes(d)
        paths.append(str(p))
    for params, n, k, seed in ((SketchParams.mash(100_000, 500, False, 31, 0), 500, 31, 0),
                               (SketchParams.mash(2000, 2000, False, 12, 42), 2000, 12, 42),
                               (SketchParams.mash(3000, 3000, False, 21, 5), 3000, 21, 5)):
        t0, _ = H.debug_file_batch()
        res = H.sketch_files(paths, params, H.FilterParams(None), n_threads=2)
        assert H.debug_file_batch()[0] - t0 == len(paths)
        for i, d in enumerate(datas):
            _same_as_oracle(res.sketch(i), d, n, k, seed)
    # asked-for filtering keeps a FASTA file out of the groups (its sketcher is the full one)
    t0, n0 = H.debug_file_batch()
    H.sketch_files(paths[:3], SketchParams.mash(1000, 1000, False, 21, 0), H.FilterParams(True), n_threads=2)
    assert H.debug_file_batch() == (t0, n0)
