"""GPU tests of the device-side BGZF inflate (fh_bgzf.hip, fh_push_bgzf_fastq): the text the kernels produce must be
the text zlib produces -- checked through the sketch of it against the oracle, for every DEFLATE block type and code
shape zlib can be made to write -- and damage must stay loud.  Run with -m gpu."""
import ctypes as C
import os
import struct
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

FH_BGZF_LAST, FH_BGZF_MORE = 1, 2


def deflate_raw(chunk: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem_level=8) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    return co.compress(chunk) + co.flush()


def bgzf_file(data: bytes, block=65280, eof_marker=True, **kw) -> bytes:
    out = []
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof_marker else [])
    for ch in chunks:
        c = deflate_raw(ch, **kw)
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(c) + 25) + c +
                   struct.pack("<II", zlib.crc32(ch), len(ch)))
    return b"".join(out)


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


def push_members(sk, chunks, deflated, batch_members, crc_of=None, launch_every=1):
    """feed (text chunk, its raw DEFLATE bytes) pairs to fh_push_bgzf_fastq, batch_members at a time; only every
    launch_every-th push (and the last) inflates, the others hand their members over with FH_BGZF_MORE"""
    n_push = 0
    L, h = sk._L, sk._h
    bufs = (C.c_void_p * 2)()
    cap, nxt = C.c_uint64(), C.c_int()
    S.check(L.fh_text_buffers(h, bufs, C.byref(cap), C.byref(nxt)))
    slot = nxt.value
    n_total = len(chunks)
    i = 0
    while True:
        take = list(range(i, min(i + batch_members, n_total)))
        i += len(take)
        last = i >= n_total
        table = bytearray()
        body = bytearray()
        off0 = 20 * len(take)
        text = 0
        for j in take:
            pad = (-len(body)) % 4
            body += b"\0" * pad
            crc = zlib.crc32(chunks[j]) if crc_of is None else crc_of(j, chunks[j])
            table += struct.pack("<5I", off0 + len(body), len(deflated[j]), text, len(chunks[j]), crc)
            body += deflated[j]
            text += len(chunks[j])
        blob = bytes(table) + bytes(body)
        assert len(blob) <= cap.value
        C.memmove(bufs[slot], blob, len(blob))
        n_push += 1
        S.check(L.fh_push_bgzf_fastq(h, len(blob), len(take), FH_BGZF_LAST if last else (FH_BGZF_MORE if n_push % launch_every else 0)))
        slot ^= 1
        if last:
            break


def new_sketcher(size, k):
    return SketchParams.mash(size, size, True, k, 0).create_sketcher()


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
    dict(level=0),                                  # stored blocks only
    dict(level=6, strategy=zlib.Z_FIXED),           # the fixed code
    dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY),    # literals only: a distance code with nothing in it
    dict(level=6, strategy=zlib.Z_RLE),             # distance 1 runs
    dict(level=6, mem_level=1),                     # many small blocks per member
]


@pytest.mark.parametrize("mode", range(len(MODES)))
def test_device_inflate_reproduces_the_text_for_every_block_type(mode):
    kw = MODES[mode]
    text = fastq_text(6000, 100 + mode, noisy_quals=(mode % 2 == 0))
    k, size = 21, 500
    o = O.OracleSketcher(O.MASH, size, k, 0, 0.001)
    assert o.sketch_stream(text) == 2
    for block, batch, every in ((65280, 7, 1), (4001, 64, 3), (65280, 1000, 1), (20000, 5, 1000)):
        chunks = [text[i:i + block] for i in range(0, len(text), block)] + [b""]
        deflated = [deflate_raw(c, **kw) for c in chunks]
        sk = new_sketcher(size, k)
        push_members(sk, chunks, deflated, batch, launch_every=every)
        assert_is_oracle_sketch(sk, o)
        sk.close()


def test_long_codes_and_sparse_alphabets():
    """codes longer than the tables' index (a skewed byte distribution gives 11-15 bit codes) and members of one symbol"""
    rng = np.random.default_rng(5)
    # header lines whose bytes follow a steep geometric distribution: Huffman code lengths reach 15
    alphabet = np.frombuffer(bytes(range(48, 48 + 64)), np.uint8)
    p = 0.5 ** np.arange(1, 65, dtype=np.float64)
    p /= p.sum()
    recs = []
    g = S.synth_genome_host(30_000, 9)
    for i in range(3000):
        name = bytes(rng.choice(alphabet, size=int(rng.integers(20, 200)), p=p)).replace(b"@", b"a")
        st = int(rng.integers(0, len(g) - 120))
        recs.append(b"@" + name + b"\n" + g[st:st + 120].tobytes() + b"\n+\n" + bytes(rng.choice(alphabet[:40], size=120, p=p[:40] / p[:40].sum())) + b"\n")
    text = b"".join(recs)
    k, size = 25, 300
    o = O.OracleSketcher(O.MASH, size, k, 0, 0.001)
    assert o.sketch_stream(text) == 2
    for kw in (dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY), dict(level=9), dict(level=1)):
        chunks = [text[i:i + 60000] for i in range(0, len(text), 60000)]
        deflated = [deflate_raw(c, **kw) for c in chunks]
        sk = new_sketcher(size, k)
        push_members(sk, chunks, deflated, 5)
        assert_is_oracle_sketch(sk, o)
        sk.close()


def test_damage_is_reported_not_sketched():
    text = fastq_text(3000, 77)
    chunks = [text[i:i + 65280] for i in range(0, len(text), 65280)]
    good = [deflate_raw(c, level=6) for c in chunks]
    def attempt(deflated, crc_of=None, chunks_=chunks):
        sk = new_sketcher(200, 21)
        try:
            with pytest.raises(_lib.FinchHipError) as ei:
                push_members(sk, chunks_, deflated, 1000, crc_of)
            assert ei.value.code == -1 and "BGZF member" in str(ei.value)  # FH_ERR_INVALID
        finally:
            sk.close()

    # a flipped bit in the middle of every member in turn (CRC-32, size or code checks catch it)
    for j in range(0, len(chunks), 3):
        bad = list(good)
        b = bytearray(bad[j])
        b[len(b) // 2] ^= 0x10
        bad[j] = bytes(b)
        attempt(bad)
    # wrong checksum in the trailer
    attempt(good, crc_of=lambda j, c: zlib.crc32(c) ^ (1 if j == 2 else 0))
    # a member cut short / with bytes appended
    attempt([good[0][:-7]] + good[1:])
    attempt([good[0] + b"\0\0\0"] + good[1:])
    # ISIZE larger than what the stream holds
    attempt(good, chunks_=[chunks[0] + b"A"] + chunks[1:], crc_of=lambda j, c: zlib.crc32(c))
    # nothing of the above leaves the library unusable
    sk = new_sketcher(200, 21)
    push_members(sk, chunks, good, 1000)
    assert sk.finish()[0] == 200
    sk.close()


def test_bgzipped_fastq_files_take_the_device_inflate(tmp_path, monkeypatch):
    """finch_sketch_files on bgzip'd reads: same sketch as the plain file and as the oracle, and the device did the inflating;
    a file the device pass refuses (here: records with blank lines between them) is read again on the host"""
    F.debug_set(read_threads="4")
    text = fastq_text(40000, 11, rl_lo=100, rl_hi=151)
    params = SketchParams.mash(2000, 2000, True, 21, 0)
    filt = H.FilterParams(False)
    plain = tmp_path / "reads.fastq"
    plain.write_bytes(text)
    paths = [str(plain)]
    for name, kw in (("l1", dict(level=1)), ("l6", dict(level=6)), ("stored", dict(level=0)), ("small", dict(level=6, block=3000))):
        p = tmp_path / ("reads_%s.fastq.gz" % name)
        p.write_bytes(bgzf_file(text, **kw))
        paths.append(str(p))
    before = H.debug_device_inflate()
    res = H.sketch_files(paths, params, filt, n_threads=2)
    after = H.debug_device_inflate()
    assert after[0] - before[0] == 4 and after[1] == before[1]
    o = O.OracleSketcher(O.MASH, 2000, 21, 0, 0.001)
    assert o.sketch_stream(text) == 2
    ref = res.sketch(0)
    for i in range(1, 5):
        sk = res.sketch(i)
        assert np.array_equal(sk.arrays[0], ref.arrays[0]) and np.array_equal(sk.arrays[1], ref.arrays[1])
        assert (sk.seq_length, sk.num_valid_kmers) == (ref.seq_length, ref.num_valid_kmers) == o.total_bases_and_kmers()
    F.debug_set(device_inflate="0")
    res0 = H.sketch_files(paths[1:2], params, filt)
    assert H.debug_device_inflate() == after
    assert np.array_equal(res0.sketch(0).arrays[0], ref.arrays[0])
    F.debug_set(device_inflate=None)
    # blank lines between records: needletail accepts them, the device splitter does not -> host parser, same answer as the oracle
    loose = text.replace(b"\n@read7/", b"\n\n@read7/")
    assert loose != text
    p = tmp_path / "loose.fastq.gz"
    p.write_bytes(bgzf_file(loose, level=1))
    r2 = H.sketch_files([str(p)], params, filt)
    assert H.debug_device_inflate()[1] == after[1] + 1
    o2 = O.OracleSketcher(O.MASH, 2000, 21, 0, 0.001)
    assert o2.sketch_stream(loose) == 2
    assert (r2.sketch(0).seq_length, r2.sketch(0).num_valid_kmers) == o2.total_bases_and_kmers()
    # damage: the host-side inflate's error is the one the caller sees
    z = bytearray(bgzf_file(text, level=6))
    z[len(z) // 2] ^= 0x55
    p = tmp_path / "damaged.fastq.gz"
    p.write_bytes(bytes(z))
    with pytest.raises(FinchError):
        H.sketch_files([str(p)], params, filt)
    # long reads: a record of several hundred kilobases spans members and batches
    rng = np.random.default_rng(1)
    g = S.synth_genome_host(3_000_000, 4)
    long_recs = []
    for i in range(12):
        rl = int(rng.integers(200_000, 900_000))
        st = int(rng.integers(0, len(g) - rl))
        long_recs.append(b"@long%d\n%s\n+\n%s\n" % (i, g[st:st + rl].tobytes(), bytes(rng.integers(40, 60, size=rl, dtype=np.uint8))))
    lt = b"".join(long_recs)
    p = tmp_path / "long.fastq.gz"
    p.write_bytes(bgzf_file(lt, level=1))
    before = H.debug_device_inflate()
    r3 = H.sketch_files([str(p)], SketchParams.mash(1000, 1000, True, 21, 0), H.FilterParams(False))
    assert H.debug_device_inflate()[0] == before[0] + 1
    o3 = O.OracleSketcher(O.MASH, 1000, 21, 0, 0.001)
    assert o3.sketch_stream(lt) == 2
    oh, ok = o3.to_vec()
    assert np.array_equal(r3.sketch(0).arrays[0], oh) and np.array_equal(r3.sketch(0).arrays[1], ok)


def test_plain_gzip_with_read_threads_goes_through_the_device_splitter_and_can_fall_back(tmp_path, monkeypatch):
    """plain gzip decoded by several threads (fh_pargz.h) feeds the device-side FASTQ splitter like any text; a file the
    splitter refuses is rewound -- through the parallel reader -- and parsed on the host"""
    import gzip
    F.debug_set(read_threads="4")
    F.debug_set(pargz_chunk="200000")
    text = fastq_text(30000, 5, rl_lo=100, rl_hi=151)
    params = SketchParams.mash(2000, 2000, True, 21, 0)
    filt = H.FilterParams(False)
    loose = text.replace(b"\n@read7/", b"\n\n@read7/")
    assert loose != text
    multi = gzip.compress(text[:len(text) // 2], 6) + gzip.compress(text[len(text) // 2:], 1)
    files = {"a.fastq.gz": gzip.compress(text, 6), "loose.fastq.gz": gzip.compress(loose, 6), "multi.fastq.gz": multi}
    paths = []
    for name, img in files.items():
        p = tmp_path / name
        p.write_bytes(img)
        paths.append(str(p))
    res = H.sketch_files(paths, params, filt, n_threads=1)
    for i, t in enumerate((text, loose, text)):
        o = O.OracleSketcher(O.MASH, 2000, 21, 0, 0.001)
        assert o.sketch_stream(t) == 2
        okc, okm = o.to_vec()
        sk = res.sketch(i)
        assert np.array_equal(sk.arrays[0], okc) and np.array_equal(sk.arrays[1], okm), paths[i]
        assert (sk.seq_length, sk.num_valid_kmers) == o.total_bases_and_kmers()


def test_the_one_symbol_at_a_time_loop_still_decodes():
    """option bgzf_serial (read once per process, hence the child): the A/B variant of the inflate kernel's symbol loop"""
    import subprocess
    import sys
    if "bgzf_serial" in os.environ.get("FH_DEBUG", ""):
        pytest.skip("this is the child")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_bgzf_device.py"), "-q", "-x", "-m", "gpu", "-k",
                        "every_block_type or long_codes or damage"], env=F.debug_env(bgzf_serial="1"), cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
