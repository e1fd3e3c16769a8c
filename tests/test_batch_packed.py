"""The batch sketcher's two-bit input form (include/finch_hip.h fh_batch_submit_packed, csrc/fh_pack2.h).

CPU part: fh_batch_pack -- host code of the library, no device -- against a numpy restatement of the classification the
sketch kernel's phase A applies (fh_core.h classify4 = needletail normalize(false) + canonical_kmers: ACGT, acgt, U/u are
bases, every other byte breaks k-mers), for every byte value, every tail length and both of its forms (AVX2, portable).
GPU part (`-m gpu`): a batch staged in the two-bit form gives the oracle's sketches bit for bit, the same files taken as
with bytes on the link."""
import ctypes as C

import numpy as np
import pytest

import finch_rs_amd as F
from finch_rs_amd import _lib

TILE, TILE_BYTES = 2048, 768
EXPECT = np.array([0xFF, ord("A"), 0xFF, ord("C"), ord("T"), ord("U"), 0xFF, ord("G")], np.uint8)
CODE = np.array([0, 0, 0, 1, 3, 3, 0, 2], np.uint8)


def region_model(stream: np.ndarray) -> np.ndarray:
    n = len(stream)
    n_tiles = (n + TILE - 1) // TILE
    pad = np.zeros(n_tiles * TILE, np.uint8)
    pad[:n] = stream
    good = ((pad & 0xDF) == EXPECT[pad & 7])
    good[n:] = False
    code = CODE[pad & 7].astype(np.uint64)
    code[n:] = 0
    out = np.zeros((n_tiles + 1) * TILE_BYTES, np.uint8)
    for t in range(n_tiles):
        c = code[t * TILE:(t + 1) * TILE].reshape(64, 32)
        words = (c << (2 * np.arange(32, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)
        g = good[t * TILE:(t + 1) * TILE].reshape(64, 32)
        gw = (g.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
        out[t * TILE_BYTES:t * TILE_BYTES + 512] = words.astype("<u8").view(np.uint8)
        out[t * TILE_BYTES + 512:(t + 1) * TILE_BYTES] = gw.astype("<u4").view(np.uint8)
    return out


def masked(region: np.ndarray) -> np.ndarray:
    """a region with the codes of positions that are no base cleared: those are unspecified (the kernel never looks at them)"""
    r = region.reshape(-1, TILE_BYTES).copy()
    codes = r[:, :512].copy().view("<u8")
    good = r[:, 512:].copy().view("<u4").astype(np.uint64)
    spread = np.zeros_like(codes)
    for i in range(32):
        spread |= ((good >> np.uint64(i)) & np.uint64(1)) * np.uint64(3 << (2 * i))
    r[:, :512] = (codes & spread).view(np.uint8)
    return r.reshape(-1)


def pack(stream: np.ndarray, slack: int = 0) -> np.ndarray:
    L = _lib.load()
    need = int(L.fh_batch_packed_bytes(len(stream)))
    region = np.full(need + slack, 0xA5, np.uint8)
    rc = L.fh_batch_pack(stream.ctypes.data if len(stream) else None, len(stream), region.ctypes.data, need)
    assert rc == 0, L.fh_last_error()
    assert np.all(region[need:] == 0xA5)  # nothing written behind what it said it needs
    return region[:need]


@pytest.mark.parametrize("scalar", [False, True])
def test_pack_equals_the_classification_model(scalar):
    rng = np.random.default_rng(77)
    F.debug_set(pack_scalar="1" if scalar else None)  # (tests/conftest.py puts FH_DEBUG back after the test)
    if True:
        # every byte value next to every other in one stream
        allb = np.concatenate([np.arange(256, dtype=np.uint8), rng.integers(0, 256, 4096, dtype=np.uint8)])
        assert np.array_equal(pack(allb, 64), region_model(allb))
        for n in [0, 1, 31, 32, 33, 63, 64, 2047, 2048, 2049, 4096, 4097, 3 * 2048 + 1000, 20000]:
            s = rng.choice(np.frombuffer(b"ACGTacgtUuNn\0-*RYKM", dtype=np.uint8), size=n)
            r = pack(s, 64)
            assert len(r) == ((n + TILE - 1) // TILE + 1) * TILE_BYTES
            assert np.array_equal(r, region_model(s)), n


def test_pack_refuses_a_region_that_is_too_small():
    L = _lib.load()
    s = np.frombuffer(b"ACGT" * 1000, dtype=np.uint8)
    region = np.zeros(10000, np.uint8)
    assert L.fh_batch_pack(s.ctypes.data, len(s), region.ctypes.data, 768 * 2) != 0
    assert L.fh_batch_pack(s.ctypes.data, len(s), region.ctypes.data, 768 * 3) == 0


def fasta_restated(text: bytes):
    """parse_fastx's FASTA records as the workers stage them: a record starts at a line that begins with '>', its sequence
    region runs to the next such line, blanks are dropped (mash.rs:73), one trailing line end is not counted (mash.rs:72)"""
    starts = [0] + [i + 1 for i in range(len(text) - 1) if text[i] == 10 and text[i + 1] == ord(">")]
    stream, bases = [], 0
    for a, b in zip(starts, starts[1:] + [len(text)]):
        rec = text[a:b]
        nl = rec.find(b"\n")
        seq = rec[nl + 1:] if nl >= 0 else b""
        trim = 0
        if seq.endswith(b"\r\n"):
            trim = 2
        elif seq.endswith(b"\n") or seq.endswith(b"\r"):
            trim = 1
        bases += len(seq) - trim
        stream.append(bytes(c for c in seq if c not in b" \t\r\n") + b"\0")
    return b"".join(stream), len(starts), bases


def _fasta_text(rng, n_rec, eol, last_eol, width):
    out = []
    for r in range(n_rec):
        L = int(rng.integers(0, 6000))
        w = np.array([20, 20, 20, 20, 3, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1] + [0.3] * 8, float)
        seq = bytes(rng.choice(np.frombuffer(b"ACGTacgtNnUuRY>-" + b" \t\xc1\xff\x7f\x00\x8a\x0b", np.uint8), size=L, p=w / w.sum()))
        hdr = b">rec%d some > description" % r
        body = eol.join(seq[j:j + width] for j in range(0, len(seq), width))
        out.append(hdr + eol + body)
    return eol.join(out) + (eol if last_eol else b"")


@pytest.mark.parametrize("form", [0, 1, 2])
@pytest.mark.parametrize("eol", [b"\n", b"\r\n"])
@pytest.mark.parametrize("piece", [1, 2, 3, 31, 64, 1000, 4099, 16384, 1 << 20])
def test_fasta_walk_in_pieces_matches_the_restatement(eol, piece, form):
    from finch_rs_amd import host as H
    F.debug_set(pack_scalar=str(form) if form else None)
    rng = np.random.default_rng(piece + len(eol))
    texts = [b">only a header", b">h" + eol, b">h" + eol + b"ACGT", b">h" + eol + b"ACGT" + eol, b">a" + eol + b">b" + eol + b"AC" + eol + eol + b">c",
             b">x" + eol + b"AC>GT" + eol + b">" + eol + b"GG\r"]
    for n_rec, last_eol, width in ((1, True, 70), (5, False, 60), (9, True, 1), (3, True, 100000), (40, False, 33)):
        texts.append(_fasta_text(rng, n_rec, eol, last_eol, width))
    for t in texts:
        if piece < 31 and len(t) > 20000:
            t = t[:20000]
        want, wrec, wbases = fasta_restated(t)
        region, m, nrec, bases = H.fasta_two_bit_probe(t, piece)
        assert (m, nrec, bases) == (len(want), wrec, wbases), (t[:40], piece)
        if form == 0:  # ... and seq_length is what the oracle's parse_fastx restatement counts (lib.rs:51-94, mash.rs:72)
            from oracle import oracle as O
            o = O.OracleSketcher(O.MASH, 10, 3, 0)
            o.sketch_stream(t)
            assert o.total_bases_and_kmers()[0] == bases, (t[:40], piece)
            # ... and the base bits of the region give the oracle's count of valid 3-mers (mash.rs:35: windows without a breaker)
            good = np.unpackbits(region.reshape(-1, TILE_BYTES)[:, 512:].copy().reshape(-1), bitorder="little")[:m].astype(np.int64)
            runs = good[:-2] + good[1:-1] + good[2:] if m >= 3 else np.zeros(0, np.int64)
            assert int((runs == 3).sum()) == o.total_bases_and_kmers()[1], (t[:40], piece)
        assert np.array_equal(masked(region), masked(region_model(np.frombuffer(want, np.uint8)))), (t[:40], piece, form)


# ---------------------------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
@pytest.mark.parametrize("k,n,seed", [(21, 1000, 0), (22, 1000, 9), (31, 1000, 0), (32, 3000, 7), (11, 100, 0), (16, 500, 0), (27, 2000, 3)])
def test_two_bit_batches_match_the_oracle(k, n, seed):
    from tests.test_gpu_batch import genome_block, same
    rng = np.random.default_rng(k * 7 + n + seed)
    lens = [int(x) for x in rng.integers(100_000, 700_000, size=9)] + [1_500_000, 65_536, 2048 * 5, 2048 * 5 + 1, 2048 * 5 - 1, 40, 0]
    blocks = [genome_block(rng, L, n_records=int(rng.integers(1, 6)), p_lower=0.02) if L else np.zeros(0, np.uint8) for L in lens]
    # other letters the reference's parser meets: U, IUPAC codes, bytes >= 0x80
    blocks[0][1000:1010] = np.frombuffer(b"UuRYKMSWBD", dtype=np.uint8)
    blocks[1][::5001] = 0xC1
    b = F.BatchSketcher(n, k, seed, max_files=8, stage_bytes=8 << 20)
    res2 = b.sketch_many(blocks, two_bit=True)
    res1 = b.sketch_many(blocks, slot=1)
    assert [r is None for r in res1] == [r is None for r in res2]
    taken = 0
    for i, (r, blk) in enumerate(zip(res2, blocks)):
        if r is None:
            continue
        taken += 1
        same(r, blk, n, k, seed, "file %d (%d bytes)" % (i, len(blk)))
        for a, c in zip(r[:3], res1[i][:3]):
            assert np.array_equal(a, c), i  # first positions too
    assert taken >= len(blocks) - 3
    b.close()


@pytest.mark.gpu
def test_two_bit_submit_checks_its_ranges():
    b = F.BatchSketcher(100, 21, 0, max_files=4, stage_bytes=1 << 20)
    with pytest.raises(F.FinchHipError):
        b.submit(0, [16], [1000], two_bit=True)  # not 64-byte aligned
    with pytest.raises(F.FinchHipError):
        b.submit(0, [0, 768], [3000, 100], two_bit=True)  # 3000 positions take 3 x 768 bytes: the second file overlaps
    with pytest.raises(F.FinchHipError):
        b.submit(0, [0], [3 << 20], two_bit=True)  # does not fit the staging buffer
    b.close()
