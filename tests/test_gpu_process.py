"""fh_process -- SketchScheme::process (mash.rs:67-80) as ONE library call per record: the record's raw sequence() bytes are
copied once, blanks dropped on the way, into the pinned staging buffer; the library puts the breaker behind them and commits
full buffers itself.  Against the oracle's process() on the same records: whitespace inside records, empty records, records
longer than the staging buffer (FH_PUSH_CONTINUE inside the library), mixed with the other ways of feeding a sketcher, and
fh_push_block's multi-threaded strip of large blocks.  Needs a real MI355X: run with `-m gpu`."""
import ctypes as C
import os

import numpy as np
import pytest

import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _same(sk, ora, ctx=""):
    kc, km, _ = sk.to_arrays()
    okc, okm = ora.to_vec()
    assert len(kc) == len(okc), (ctx, len(kc), len(okc))
    for f in ("hash", "count", "extra_count"):
        assert np.array_equal(kc[f], okc[f]), (ctx, f)
    assert np.array_equal(km, okm), ctx
    tb = C.c_uint64()
    F._lib.check(sk._L.fh_total_bases(sk._h, C.byref(tb)))
    assert (tb.value, sk.finish()[1]) == ora.total_bases_and_kmers(), ctx


def _records(rng, n, lo, hi, p_blank=0.02):
    recs = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        r = rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), size=L, p=[.22, .22, .22, .22, .02, .02, .02, .02, .04])
        m = rng.random(L)
        r[m < p_blank] = rng.choice(np.frombuffer(b"\n\r \t", dtype=np.uint8), size=int((m < p_blank).sum()))
        recs.append(bytes(r))
    return recs


@pytest.mark.parametrize("k,n", [(21, 1000), (31, 50), (40, 1000)])
def test_one_call_per_record_matches_the_oracle(k, n):
    rng = np.random.default_rng(k)
    recs = _records(rng, 3000, 0, 400) + [b"", b"\n", b"ACGT" * 2000 + b"\n" + b"TTGCA" * 900]
    sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher()
    ora = O.OracleSketcher(O.MASH, n, k, 0)
    for r in recs:
        sk.process(r)
        ora.process(r)
    _same(sk, ora)
    sk.close()


def test_records_in_one_call_and_scaled():
    rng = np.random.default_rng(9)
    recs = _records(rng, 20000, 80, 260, p_blank=0.01)
    base = np.frombuffer(b"#".join(recs), dtype=np.uint8)  # ('#': bytes between the records that belong to none)
    lens = np.array([len(r) for r in recs], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens[:-1] + 1)]).astype(np.uint64)
    sk = F.SketchParams.scaled(500, 21, 0.002, 0).create_sketcher()
    ora = O.OracleSketcher(O.SCALED, 500, 21, 0, scale=0.002)
    sk.process_records(base, offs, lens)
    for r in recs:
        ora.process(r)
    _same(sk, ora)
    sk.close()


def test_a_record_outside_the_buffer_is_refused_where_the_loop_is():
    """bounds are checked per record in the library (no sum of offset and length: it wraps); the records in front of the
    offending one have been taken, and the sketcher goes on working"""
    rng = np.random.default_rng(10)
    recs = _records(rng, 300, 80, 260)
    base = np.frombuffer(b"#".join(recs), dtype=np.uint8)
    lens = np.array([len(r) for r in recs], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens[:-1] + 1)]).astype(np.uint64)
    for bad_off, bad_len in ((base.size - 10, 11), (base.size + 1, 0), (2**64 - 5, 10), (5, 2**64 - 3)):
        sk = F.SketchParams.default().create_sketcher()
        o2, l2 = offs.copy(), lens.copy()
        o2[200], l2[200] = bad_off, bad_len
        with pytest.raises(ValueError):
            sk.process_records(base, o2, l2)
        assert sk.total_bases == int(lens[:200].sum())
        sk.process_records(base, offs[200:], lens[200:])
        ora = O.OracleSketcher(O.MASH, 1000, 21, 0)
        for r in recs:
            ora.process(r)
        _same(sk, ora)
        sk.close()
    with pytest.raises(ValueError):
        F.SketchParams.default().create_sketcher().process_records(base, offs, lens[:-1])


def test_records_longer_than_the_staging_buffer(monkeypatch):
    """a 300 kb record through 64 KiB staging buffers: the library cuts it and continues (k-mers span the cuts)"""
    F.debug_set(stage_bytes="65536")
    F.load().fh_release_cached()
    rng = np.random.default_rng(4)
    recs = _records(rng, 5, 250_000, 300_000, p_blank=0.015) + _records(rng, 300, 10, 500)
    sk = F.SketchParams.default().create_sketcher()
    ora = O.OracleSketcher(O.MASH, 1000, 21, 0)
    for r in recs:
        sk.process(r)
        ora.process(r)
    _same(sk, ora)
    sk.close()
    F.load().fh_release_cached()


def test_mixed_with_blocks_and_a_second_run_on_the_same_handle():
    rng = np.random.default_rng(2)
    a, b, c = _records(rng, 500, 50, 300), _records(rng, 500, 50, 300), _records(rng, 500, 50, 300)
    sk = F.SketchParams.default().create_sketcher()
    for rep in range(2):
        sk.reset()
        ora = O.OracleSketcher(O.MASH, 1000, 21, 0)
        for r in a:
            sk.process(r)
        sk.push_block(b"\0".join(b) + b"\0")  # (commits the waiting records first)
        for r in c:
            sk.process(r)
        for r in a + b + c:
            ora.process(r)
        kc, km, _ = sk.to_arrays()
        okc, okm = ora.to_vec()
        assert np.array_equal(kc, okc) and np.array_equal(km, okm)
        assert sk.finish()[1] == ora.total_bases_and_kmers()[1]
    sk.close()


def test_push_block_strips_a_large_block_on_several_threads():
    """a 24 MB block with line breaks inside its records (multi-line FASTA records as a caller might hand them over): the strip
    of fh_push_block runs on several threads above 4 MB -- same sketch as the record-by-record oracle"""
    g = S.synth_genome_host(24_000_000, 21)
    lines = g.reshape(-1, 60)
    text = np.concatenate([lines, np.full((lines.shape[0], 1), 10, np.uint8)], axis=1).reshape(-1)
    cuts = [0, 5_000_011, 5_000_012, 17_000_000, len(text)]
    block = b"\0".join(bytes(text[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)) + b"\0"
    sk = F.SketchParams.default().create_sketcher()
    sk.push_block(block)
    ora = O.OracleSketcher(O.MASH, 1000, 21, 0)
    for i in range(len(cuts) - 1):
        ora.process(bytes(text[cuts[i]:cuts[i + 1]]))
    kc, km, _ = sk.to_arrays()
    okc, okm = ora.to_vec()
    assert np.array_equal(kc, okc) and np.array_equal(km, okm)
    assert sk.finish()[1] == ora.total_bases_and_kmers()[1]
    sk.close()
