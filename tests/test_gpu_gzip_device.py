"""GPU tests of the device-side inflate of PLAIN gzip (fh_bgzf.hip: k_gz_chunks / k_gz_chain / k_gz_win_* / k_gz_text,
fh_push_gzip_fastq): one DEFLATE stream cut into chunks, every chunk decoded from a block start found by search, the
chain of chunks stitched and the markers of the unknown windows looked up.  The text must be the text zlib produces --
checked through the sketch of it against the oracle and through the stream's own CRC-32 -- for every block type, for
chunks far smaller than a block (most then hold no start at all), across batches (undecoded bytes, window and partial
record carried over), for batches handed over in pieces while the launch that decodes them is already waiting, and
anything the device pass cannot vouch for must end up with the host-side inflate's verdict.  Run with -m gpu."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

import finch_rs_amd as F

from finch_rs_amd import _lib
from finch_rs_amd import host as H
from finch_rs_amd import sketch_schemes as S
from finch_rs_amd.sketch_schemes import FinchError, SketchParams
from oracle import oracle as O

pytestmark = pytest.mark.gpu

FH_GZ_FIRST, FH_GZ_LAST, FH_GZ_MORE = 1, 2, 4
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def deflate_raw(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem_level=8, flush_every=0) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    if not flush_every:
        return co.compress(data) + co.flush()
    out = []
    for i in range(0, len(data), flush_every):  # Z_FULL_FLUSH: an empty stored block and a fresh window in mid-stream
        out.append(co.compress(data[i:i + flush_every]))
        out.append(co.flush(zlib.Z_FULL_FLUSH if (i // flush_every) % 2 else zlib.Z_SYNC_FLUSH))
    out.append(co.flush())
    return b"".join(out)


def gzip_file(data: bytes, name=b"", **kw) -> bytes:
    flg = 8 if name else 0
    return (b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\x00\xff" + (name + b"\0" if name else b"") + deflate_raw(data, **kw) +
            (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little"))


def fastq_text(n_reads, seed, rl_lo=30, rl_hi=300, noisy_quals=True):
    rng = np.random.default_rng(seed)
    g = S.synth_genome_host(50_000, seed)
    recs = []
    for i in range(n_reads):
        rl = int(rng.integers(rl_lo, rl_hi + 1))
        st = int(rng.integers(0, len(g) - rl))
        seq = bytearray(g[st:st + rl].tobytes())
        if rng.random() < 0.2:
            seq[int(rng.integers(0, rl))] = ord("N")
        q = bytes(rng.integers(35, 74, size=rl, dtype=np.uint8)) if noisy_quals else b"I" * rl
        recs.append(b"@read%d/%d\n%s\n+\n%s\n" % (i, seed, bytes(seq), q))
    return b"".join(recs)


def push_stream(sk, body: bytes, push_bytes=None, piece_bytes=None):
    """the DEFLATE bytes of one member (trailer included) through fh_push_gzip_fastq: batches of push_bytes, each handed over
    in pieces of piece_bytes (FH_GZ_MORE on all but the last piece of a batch)"""
    L, h = sk._L, sk._h
    bufs = (C.c_void_p * 2)()
    cap, nxt, bcap = C.c_uint64(), C.c_int(), C.c_uint64()
    S.check(L.fh_text_buffers(h, bufs, C.byref(cap), C.byref(nxt)))
    S.check(L.fh_gzip_batch_capacity(h, C.byref(bcap)))
    slot = nxt.value
    step = min(push_bytes or bcap.value, bcap.value, cap.value)
    done, trailing = C.c_uint32(), C.c_uint64()
    n_push = 0
    for o in range(0, max(1, len(body)), step):
        batch = body[o:o + step]
        last = o + step >= len(body)
        C.memmove(bufs[slot], batch, len(batch))
        piece = piece_bytes or max(1, len(batch))
        offs = list(range(0, max(1, len(batch)), piece))
        for i, po in enumerate(offs):
            n = min(piece, len(batch) - po)
            more = i + 1 < len(offs)
            flags = (FH_GZ_FIRST if o == 0 and i == 0 else 0) | (FH_GZ_MORE if more else (FH_GZ_LAST if last else 0))
            S.check(L.fh_push_gzip_fastq(h, n, flags, C.byref(done), C.byref(trailing)))
            n_push += 1
        slot ^= 1
        if done.value:
            break
    return done.value, trailing.value, n_push


def new_sketcher(size, k, **kw):
    return SketchParams.mash(size, size, True, k, 0).create_sketcher(**kw)


def assert_is_oracle_sketch(sk, o):
    kc, km, _ = sk.to_arrays()
    okc, okm = o.to_vec()
    assert np.array_equal(kc, okc) and np.array_equal(km, okm)
    tb = C.c_uint64()
    S.check(sk._L.fh_text_bases(sk._h, C.byref(tb)))
    assert (tb.value, sk.finish()[1]) == o.total_bases_and_kmers()


MODES = [
    dict(level=6),                                  # dynamic codes, typical
    dict(level=1),                                  # what most pipelines write
    dict(level=9, mem_level=9),
    dict(level=0),                                  # stored blocks only: no start to be found anywhere
    dict(level=6, strategy=zlib.Z_FIXED),           # the fixed code: likewise
    dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY),    # literals only: a distance code with nothing in it
    dict(level=6, strategy=zlib.Z_RLE),             # distance 1 runs
    dict(level=6, mem_level=1),                     # many small blocks
    dict(level=6, flush_every=50_000),              # empty stored blocks and window resets in mid-stream
    dict(level=4, mem_level=3, flush_every=7_001),
]


@pytest.fixture
def chunk_env():
    yield
    F.debug_set(gz_chunk=None)


@pytest.mark.parametrize("mode", range(len(MODES)))
def test_chunked_inflate_reproduces_the_text_for_every_block_type(mode, chunk_env):
    kw = MODES[mode]
    text = fastq_text(8000, 300 + mode, noisy_quals=(mode % 2 == 0))
    k, size = 21, 500
    o = O.OracleSketcher(O.MASH, size, k, 0, 0.001)
    assert o.sketch_stream(text) == 2
    body = deflate_raw(text, **kw) + zlib.crc32(text).to_bytes(4, "little") + len(text).to_bytes(4, "little")
    for chunk in (None, 1024, 4096, 20000, 100_000):  # None: the library's choice (one chunk for an input this small)
        if chunk is None:
            F.debug_set(gz_chunk=None)
        else:
            F.debug_set(gz_chunk=str(chunk))
        sk = new_sketcher(size, k)
        done, trailing, _ = push_stream(sk, body)
        assert (done, trailing) == (1, 0), (kw, chunk)
        assert_is_oracle_sketch(sk, o)
        sk.close()


@pytest.mark.parametrize("level,push", [(1, 1 << 20), (6, 1 << 20), (6, 300_000), (9, 2_000_000)])
def test_undecoded_bytes_window_and_partial_record_carry_over_between_pushes(level, push, chunk_env):
    text = b"".join(fastq_text(9000, 400 + i) for i in range(4))
    k, size = 21, 1000
    o = O.OracleSketcher(O.MASH, size, k, 0, 0.001)
    assert o.sketch_stream(text) == 2
    body = deflate_raw(text, level=level) + zlib.crc32(text).to_bytes(4, "little") + len(text).to_bytes(4, "little")
    assert len(body) > 3 * push
    F.debug_set(gz_chunk="65536")
    sk = new_sketcher(size, k, stage_bytes=8 << 20)
    done, trailing, n_push = push_stream(sk, body, push)
    assert (done, trailing) == (1, 0) and n_push >= 3
    assert_is_oracle_sketch(sk, o)
    # the handle again, after a reset
    sk.reset()
    done, trailing, _ = push_stream(sk, body, push * 2)
    assert (done, trailing) == (1, 0)
    assert_is_oracle_sketch(sk, o)
    sk.close()


@pytest.mark.parametrize("level,piece", [(1, 1 << 20), (6, 1_300_000), (6, 3 << 20), (9, 200_000)])
def test_a_batch_handed_over_in_pieces_is_decoded_while_it_comes_in(level, piece, chunk_env):
    """FH_GZ_MORE: the chunks in front of the newest piece are launched with what is there; the verdicts of a chunk on where
    it stops and of the chunk that begins there must agree whatever was there when either ran"""
    text = b"".join(fastq_text(9000, 500 + i) for i in range(8))
    k, size = 21, 1000
    o = O.OracleSketcher(O.MASH, size, k, 0, 0.001)
    assert o.sketch_stream(text) == 2
    body = deflate_raw(text, level=level) + zlib.crc32(text).to_bytes(4, "little") + len(text).to_bytes(4, "little")
    for chunk in ("8192", None):
        if chunk:
            F.debug_set(gz_chunk=chunk)
        else:
            F.debug_set(gz_chunk=None)
        sk = new_sketcher(size, k)
        done, trailing, n_push = push_stream(sk, body, None, piece)
        assert (done, trailing) == (1, 0) and n_push >= 3
        assert_is_oracle_sketch(sk, o)
        sk.close()


def test_trailing_bytes_are_reported_and_damage_is_loud(chunk_env):
    text = fastq_text(5000, 77)
    body = deflate_raw(text, level=6) + zlib.crc32(text).to_bytes(4, "little") + len(text).to_bytes(4, "little")
    F.debug_set(gz_chunk="16384")
    sk = new_sketcher(500, 21)
    assert push_stream(sk, body + b"x" * 37)[:2] == (1, 37)
    sk.reset()
    # a wrong CRC-32, a wrong size, a flipped bit in the middle, a stream cut short
    bad_crc = body[:-8] + bytes([body[-8] ^ 1]) + body[-7:]
    bad_len = body[:-1] + bytes([body[-1] ^ 0x40])
    flipped = body[:len(body) // 2] + bytes([body[len(body) // 2] ^ 0x10]) + body[len(body) // 2 + 1:]
    for damaged in (bad_crc, bad_len, flipped, body[:len(body) // 3], body[:-9]):
        with pytest.raises(_lib.FinchHipError):
            push_stream(sk, damaged)
        sk.reset()
    # ... and the handle is none the worse for it
    o = O.OracleSketcher(O.MASH, 500, 21, 0, 0.001)
    o.sketch_stream(text)
    assert push_stream(sk, body)[:2] == (1, 0)
    assert_is_oracle_sketch(sk, o)
    sk.close()


def test_a_batch_abandoned_half_way_does_not_leave_the_device_waiting(chunk_env):
    """FH_GZ_MORE launches the batch's decoding at once; if the rest never comes, fh_reset (and fh_free) tell the launch to give up
    -- promptly, not after its three-second patience -- and the handle decodes the next stream as if nothing had happened"""
    import time
    text = b"".join(fastq_text(9000, 700 + i) for i in range(3))
    body = deflate_raw(text, level=6) + zlib.crc32(text).to_bytes(4, "little") + len(text).to_bytes(4, "little")
    o = O.OracleSketcher(O.MASH, 500, 21, 0, 0.001)
    o.sketch_stream(text)
    sk = new_sketcher(500, 21)
    L, h = sk._L, sk._h
    bufs = (C.c_void_p * 2)()
    cap, nxt = C.c_uint64(), C.c_int()
    S.check(L.fh_text_buffers(h, bufs, C.byref(cap), C.byref(nxt)))
    done, trailing = C.c_uint32(), C.c_uint64()
    for _ in range(2):
        piece = body[:len(body) // 3]
        C.memmove(bufs[nxt.value], piece, len(piece))
        S.check(L.fh_push_gzip_fastq(h, len(piece), FH_GZ_FIRST | FH_GZ_MORE, C.byref(done), C.byref(trailing)))
        t0 = time.perf_counter()
        sk.reset()
        assert time.perf_counter() - t0 < 1.0
        S.check(L.fh_text_buffers(h, bufs, C.byref(cap), C.byref(nxt)))
    assert push_stream(sk, body, None, 1 << 20)[:2] == (1, 0)
    assert_is_oracle_sketch(sk, o)
    # ... and a handle freed in that state
    sk2 = new_sketcher(500, 21)
    S.check(sk2._L.fh_text_buffers(sk2._h, bufs, C.byref(cap), C.byref(nxt)))
    C.memmove(bufs[nxt.value], body[:100_000], 100_000)
    S.check(sk2._L.fh_push_gzip_fastq(sk2._h, 100_000, FH_GZ_FIRST | FH_GZ_MORE, C.byref(done), C.byref(trailing)))
    t0 = time.perf_counter()
    sk2.close()
    assert time.perf_counter() - t0 < 1.0
    sk.close()


def sketch_of(path, p, device_gzip=True, **kw):
    if not device_gzip:
        F.debug_set(device_gzip="0")
    try:
        sk = H.sketch_files([path], p, H.FilterParams(False), **kw).sketch(0)
        return sk.arrays[0].tobytes(), sk.arrays[1].tobytes(), sk.seq_length, sk.num_valid_kmers
    finally:
        F.debug_set(device_gzip=None)


def test_sketch_files_takes_gzip_through_the_device_and_falls_back_when_it_must(tmp_path, chunk_env):
    p = SketchParams.mash(1000, 1000, True, 21, 0)
    text = fastq_text(20000, 11)
    o = O.OracleSketcher(O.MASH, 1000, 21, 0, 0.001)
    o.sketch_stream(text)
    okc, okm = o.to_vec()
    want = (okc.tobytes(), okm.tobytes()) + o.total_bases_and_kmers()
    F.debug_set(gz_chunk="32768")
    cases = {
        "plain.fastq.gz": (gzip.compress(text, 6), 1, 0),
        "named.fastq.gz": (gzip_file(text, name=b"reads.fastq", level=1), 1, 0),
        "stored.fastq.gz": (gzip_file(text, level=0), 1, 0),
        # two members, bytes behind the trailer: the host-side reader's business
        "two.fastq.gz": (gzip.compress(text[:len(text) // 2 + 17], 6) + gzip.compress(text[len(text) // 2 + 17:], 6), 1, 1),
    }
    for name, (img, dev, reread) in cases.items():
        path = str(tmp_path / name)
        open(path, "wb").write(img)
        before = H.debug_device_gzip()
        got = sketch_of(path, p, n_threads=4)
        after = H.debug_device_gzip()
        assert got == want, name
        assert (after[0] - before[0] + after[1] - before[1], after[1] - before[1]) == (dev, reread), name
        assert sketch_of(path, p, device_gzip=False, n_threads=4) == want, name
    # FASTA text is not this path's (the probe of the first byte says so), and neither is a file that is no gzip stream at all
    fa = b">g\n" + S.synth_genome_host(300_000, 5).tobytes() + b"\n"
    path = str(tmp_path / "g.fa.gz")
    open(path, "wb").write(gzip.compress(fa, 6))
    before = H.debug_device_gzip()
    a = sketch_of(path, p, n_threads=4)
    assert H.debug_device_gzip() == before
    assert a == sketch_of(path, p, device_gzip=False, n_threads=4)
    # damage: the same refusal either way
    img = bytearray(gzip.compress(text, 6))
    img[len(img) // 2] ^= 0x20
    path = str(tmp_path / "damaged.fastq.gz")
    open(path, "wb").write(bytes(img))
    errs = []
    for dev in (True, False):
        with pytest.raises(FinchError) as e:
            sketch_of(path, p, device_gzip=dev, n_threads=4)
        errs.append(str(e.value))
    assert errs[0] == errs[1]
    path = str(tmp_path / "short.fastq.gz")
    open(path, "wb").write(gzip.compress(text, 6)[:-20000])
    for dev in (True, False):
        with pytest.raises(FinchError):
            sketch_of(path, p, device_gzip=dev, n_threads=4)


def test_a_file_of_several_pushes_equals_the_host_side_inflate(tmp_path):
    """~85 MB of DEFLATE bytes: a short first push and full staging buffers behind it, chunk size the library's own"""
    text = b"".join(fastq_text(30000, 900 + i, rl_lo=140, rl_hi=160) for i in range(36))
    path = str(tmp_path / "big.fastq.gz")
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    open(path, "wb").write(co.compress(text) + co.flush())
    assert os.path.getsize(path) > (70 << 20)
    p = SketchParams.mash(1000, 1000, True, 21, 0)
    before = H.debug_device_gzip()
    a = sketch_of(path, p, n_threads=8)
    after = H.debug_device_gzip()
    assert (after[0] - before[0], after[1] - before[1]) == (1, 0)
    assert a == sketch_of(path, p, device_gzip=False, n_threads=8)
    assert a[2] == sum(len(l) for l in text.split(b"\n")[1::4])
