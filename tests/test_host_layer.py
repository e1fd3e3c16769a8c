"""CPU tests of the C++ host layer (include/finch_host.h): FASTX reader vs the oracle's parser, the
filters vs the reference's known answers (filtering.rs tests) and the oracle, and the `.sk` writer."""
import gzip
import json

import numpy as np
import pytest

import finch_rs_amd as F

from finch_rs_amd import host as H
from finch_rs_amd.sketch_schemes import FinchError, KC_DTYPE, SketchParams
from oracle import oracle as O


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as G
    G.build()


def oracle_scan(data: bytes):
    s = O.OracleSketcher(O.MASH, 10, 21, 0)
    fmt = s.sketch_stream(data)
    return s.total_bases_and_kmers()[0], fmt


FASTA_CASES = [
    b">a\nACGT\n",
    b">a\nACGT",
    b">a desc\nACGT\nTTGA\n>b\nAC\nGT\n\n",
    b">a\r\nACGT\r\nTT\r\n>b\r\nGG\r\n",
    b">only header\n",
    b">a\nAC GT\tNN\n>b\n\n>c\nA\n",
]


@pytest.mark.parametrize("data", FASTA_CASES)
def test_fasta_scan_matches_oracle(data):
    n, tb, fmt = H.fastx_scan(data)
    otb, ofmt = oracle_scan(data)
    assert fmt == 1 and ofmt == 1
    assert tb == otb
    assert n == data.count(b">")


def test_fastq_scan_and_errors(golden_dir):
    fq = b"@r1\nACGT\n+\nIIII\n@r2 x\nAC\n+r2\nII\n@r3\n\n+\n\n"
    n, tb, fmt = H.fastx_scan(fq)
    assert (n, tb, fmt) == (3, 6, 2)
    assert oracle_scan(fq) == (6, 2)
    n, tb, fmt = H.fastx_scan(b"@r1\r\nACGT\r\n+\r\nIIII\r\n@r2\nAC\n+\nII")  # CRLF and no final newline
    assert (n, tb) == (2, 6)
    assert H.fastx_scan(gzip.compress(fq)) == (3, 6, 2)
    import bz2, lzma
    assert H.fastx_scan(bz2.compress(fq)) == (3, 6, 2)     # needletail sniffs 'BZ' ...
    assert H.fastx_scan(lzma.compress(fq)) == (3, 6, 2)    # ... and FD 37 (xz)
    big = fq * 200000
    assert H.fastx_scan(bz2.compress(big)) == (600000, 1200000, 2)
    assert H.fastx_scan(lzma.compress(big, preset=1)) == (600000, 1200000, 2)
    assert H.fastx_scan(gzip.compress(big, 1) + gzip.compress(fq)) == (600003, 1200006, 2)  # concatenated members
    for bad in [b"@r1\nACGT\n+\nIII\n", b"@r1\nACGT\nIIII\n+\n", b"@r1\nACGT\n", b"xyz", b""]:
        with pytest.raises(FinchError):
            H.fastx_scan(bad)
    data = open(golden_dir + "/query.fa", "rb").read()
    assert H.fastx_scan(data) == (3, 134 + 136 + 135, 1)


def test_big_records_cross_buffer_refills():
    rng = np.random.default_rng(0)
    seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=20_000_000))
    lines = b"\n".join(seq[i:i + 70] for i in range(0, len(seq), 70))
    data = b">chr\n" + lines + b"\n>small\nACGT\n"
    n, tb, fmt = H.fastx_scan(data)
    assert n == 2 and fmt == 1
    assert tb == len(lines) + 4
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, seq[i * 150:(i + 1) * 150], b"I" * 150) for i in range(100000))
    assert H.fastx_scan(fq) == (100000, 15_000_000, 2)


def kc_of(counts, extras=None):
    kc = np.zeros(len(counts), dtype=KC_DTYPE)
    kc["hash"] = np.arange(1, len(counts) + 1)
    kc["count"] = counts
    kc["extra_count"] = extras if extras is not None else 0
    return kc


def test_guess_filter_threshold_reference_known_answers():
    # lib/src/filtering.rs:197-327
    cases = [([], 0.2, 1), ([1], 0.2, 1), ([1, 1], 0.2, 1), ([1, 9], 0.2, 8), ([1, 10, 10, 9], 0.1, 8),
             ([1, 1, 2, 4], 0.1, 1), ([2], 1.0, 2)]
    for counts, level, want in cases:
        assert H.guess_filter_threshold(counts, level) == want
        assert O.guess_filter_threshold(kc_of(counts), level) == want


def test_filter_counts_matches_oracle_and_updates_params():
    rng = np.random.default_rng(9)
    n = 5000
    counts = np.concatenate([rng.poisson(1.2, n // 2) + 1, rng.poisson(40, n - n // 2) + 1]).astype(np.uint32)
    extras = (counts * rng.random(n)).astype(np.uint32)
    kc = kc_of(counts, extras)
    kc["hash"] = np.sort(rng.integers(1, 2**62, n).astype(np.uint64))
    km = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, 21))
    params = SketchParams.mash(n, 1000, True, 21, 0)
    sk = H.sketches_from_arrays("x", 1, 2, kc, km, params, H.FilterParams(False))
    fp = sk.apply_filters(0, H.FilterParams(True, (None, None), 0.31, 0.1))
    got = sk.sketch(0)
    # oracle: strand -> err threshold -> abundance -> truncate(final_size)
    a, ak = O.filter_strands(kc, km, 0.1)
    cutoff = O.guess_filter_threshold(a, 0.31)
    b, bk = O.filter_abundance(a, ak, cutoff, None)
    assert fp.abun_filter == (cutoff, None) and fp.filter_on is True
    assert np.array_equal(got.arrays[0], b[:1000]) and np.array_equal(got.arrays[1], bk[:1000])
    # filtering off: untouched apart from the truncate
    sk2 = H.sketches_from_arrays("x", 1, 2, kc, km, params, H.FilterParams(False))
    sk2.apply_filters(0, H.FilterParams(False, (None, None), 0.31, 0.1))
    assert np.array_equal(sk2.sketch(0).arrays[0], kc[:1000])
    # strict: too few kmers -> the reference's error text (mod.rs:123-125)
    sk3 = H.sketches_from_arrays("reads.fq", 1, 2, kc[:10], km[:10], SketchParams.mash(100, 100, False, 21, 0), H.FilterParams(False))
    with pytest.raises(FinchError, match=r"reads.fq had too few kmers \(10\) to sketch"):
        sk3.apply_filters(0, H.FilterParams(False))


def test_records_no_sketcher_can_emit_are_rejected_at_the_abi():
    """count == 0 would index the count histogram at -1 (statistics.rs:30-47 panics there), extra_count > count would
    underflow the strand filter (filtering.rs:424): caller-supplied arrays are checked at the boundary."""
    params = SketchParams.mash(10, 10, True, 21, 0)
    km = np.full((3, 21), ord("A"), np.uint8)
    bad0 = kc_of([3, 0, 2])
    bad0["hash"] = [1, 2, 3]
    with pytest.raises(FinchError, match="count 0"):
        H.sketches_from_arrays("x", 1, 2, bad0, km, params, H.FilterParams(False))
    bad1 = kc_of([3, 1, 2], [0, 2, 0])
    bad1["hash"] = [1, 2, 3]
    with pytest.raises(FinchError, match="extra_count 2 > count 1"):
        H.sketches_from_arrays("x", 1, 2, bad1, km, params, H.FilterParams(False))
    assert H.guess_filter_threshold([4, 0, 1], 0.2) == 0  # 0 is never a threshold: error return
    assert b"is 0" in H.lib().finch_last_error()
    assert H.lib().finch_guess_filter_threshold(None, 5, 0.2) == 0
    ok = kc_of([3, 1, 2], [0, 1, 2])
    ok["hash"] = [1, 2, 3]
    sk = H.sketches_from_arrays("x", 1, 2, ok, km, params, H.FilterParams(False))
    sk.apply_filters(0, H.FilterParams(True, (None, None), 0.5, 0.1))


def test_hist_and_cardinality_reference_known_answers():
    """statistics.rs:53-129 (hist, incl. the issue-63 regression) and the k-minimum-values cardinality (8-23)"""
    p = SketchParams.mash(10, 10, True, 4, 0)
    km = np.full((5, 4), ord("A"), np.uint8)
    a = kc_of([1, 1, 1])
    sk = H.sketches_from_arrays("a", 1, 1, a, km[:3], p, H.FilterParams(False))
    assert sk.hist(0).tolist() == [3]
    b = kc_of([4, 2, 4, 3, 126497])
    b["hash"] = [1, 2, 3, 4, 5]
    sk = H.sketches_from_arrays("b", 1, 1, b, km, p, H.FilterParams(False))
    hd = sk.hist(0)
    assert len(hd) == 126497 and (hd[0], hd[1], hd[2], hd[3], hd[126496]) == (0, 1, 1, 2, 1) and hd.sum() == 5
    # cardinality: (len - 1) / (last hash / 2^64) in f32, `as u64`
    c = kc_of([1] * 4)
    c["hash"] = [10, 20, 30, 2**63]
    sk = H.sketches_from_arrays("c", 1, 1, c, km[:4], p, H.FilterParams(False))
    assert sk.cardinality(0) == int(np.float32(3) / (np.float32(2**63) / np.float32(2**64))) == 6
    e = kc_of([])
    assert H.sketches_from_arrays("e", 0, 0, e, km[:0], p, H.FilterParams(False)).cardinality(0) == 0
    z = kc_of([1, 1])
    z["hash"] = [0, 5]
    sk = H.sketches_from_arrays("z", 1, 1, z, km[:2], p, H.FilterParams(False))
    want = np.float32(1) / (np.float32(5) / np.float32(2**64))
    assert sk.cardinality(0) == min(int(want), 2**64 - 1)


def test_sk_json_writer_format():
    kc = kc_of([3, 1], [1, 0])
    kc["hash"] = [12345678901234567890, 18446744073709551615]
    km = np.frombuffer(b"ACGTA" + b"TTTTT", np.uint8).reshape(2, 5)
    p = SketchParams.mash(2, 2, False, 5, 42)
    sk = H.sketches_from_arrays('we"ird\\name\n', 100, 96, kc, km, p, H.FilterParams(True, (2, None), 0.25, 0.1))
    js = sk.to_json()
    # serialization/json.rs:64-89,141-158 field order, hashes as strings, extra_count not serialised
    assert js == ('{"kmer":5,"alphabet":"ACGT","preserveCase":false,"canonical":true,"sketchSize":2,'
                  '"hashType":"MurmurHash3_x64_128","hashBits":64,"hashSeed":42,"scale":null,"sketches":['
                  '{"name":"we\\"ird\\\\name\\n","seqLength":100,"numValidKmers":96,"comment":"",'
                  '"filters":{"strandFilter":"0.1","errFilter":"0.25","minCopies":"2"},'
                  '"hashes":["12345678901234567890","18446744073709551615"],"kmers":["ACGTA","TTTTT"],"counts":[3,1]}]}')
    d = json.loads(js)
    assert d["sketches"][0]["name"] == 'we"ird\\name\n'
    p2 = SketchParams.scaled(7, 5, 0.001, 0)
    js2 = H.sketches_from_arrays("s", 1, 1, kc, km, p2, H.FilterParams(False)).to_json()
    d2 = json.loads(js2)
    assert '"scale":0.001,' in js2 and d2["sketchSize"] == 7 and d2["sketches"][0]["filters"] == {}
    for scale, txt in [(1.0, "1.0"), (0.5, "0.5"), (1e-7, "1e-7"), (0.00123, "0.00123")]:
        assert ('"scale":%s,' % txt) in H.sketches_from_arrays("s", 1, 1, kc, km, SketchParams.scaled(7, 5, scale, 0),
                                                              H.FilterParams(False)).to_json()


def test_raw_distance_reference_known_answers():
    # lib/src/distance.rs:176-240
    assert H.raw_distance([0, 1, 2], [1, 2]) == (2. / 2., 2. / 3., 2, 3)
    assert H.raw_distance([0, 2], [1, 2]) == (1. / 2., 1. / 3., 1, 3)
    assert H.raw_distance([0, 1], [2, 3]) == (0., 0., 0, 2)
    assert H.raw_distance([], []) == (0., 1., 0, 0)
    assert H.raw_distance([], [5]) == (0., 1., 0, 0)
    # scaled: 1e-18 -> max_hash 18
    assert H.raw_distance([10, 15, 20], [15, 20], 1e-18) == (1., 2. / 3., 2, 3)
    assert H.raw_distance([5, 10, 15], [5, 10], 1e-18) == (1., 2. / 3., 2, 3)
    assert H.raw_distance([5, 10, 15, 20], [5, 10], 1e-18) == (1., 2. / 3., 2, 3)
    assert H.raw_distance([5, 10], [5, 10, 15, 20], 1e-18) == (2. / 3., 2. / 3., 2, 3)


def test_distance_between_sketches():
    # distance.rs:314-337 shape: identical sketches -> jaccard 1, containment 1, mash distance 0
    kc = kc_of([1, 1, 2], [1, 0, 1])
    kc["hash"] = [11, 22, 33]
    km = np.frombuffer(b"ccacaa", np.uint8).reshape(3, 2)
    p = SketchParams.scaled(3, 2, 0.001, 42)
    a = H.sketches_from_arrays("a", 1, 1, kc, km, p, H.FilterParams(False))
    b = H.sketches_from_arrays("b", 1, 1, kc, km, p, H.FilterParams(False))
    d = H.distance(a, 0, b, 0)
    assert (d["jaccard"], d["containment"], d["common_hashes"], d["mash_distance"]) == (1.0, 1.0, 3, 0.0)
    kc2 = kc.copy()
    kc2["hash"] = [11, 22, 44]
    c = H.sketches_from_arrays("c", 1, 1, kc2, km, SketchParams.mash(3, 3, True, 2, 42), H.FilterParams(False))
    d = H.distance(a, 0, c, 0)
    # the walk stops when one sketch is exhausted: 33 < 44 ends the query, the reference's 44 is never counted
    assert d["common_hashes"] == 2 and d["total_hashes"] == 3 and d["jaccard"] == 2. / 3.
    import math
    j = 2. / 3.
    assert abs(d["mash_distance"] - (-math.log(2 * j / (1 + j)) / 2)) < 1e-15
    assert H.distance(a, 0, c, 0, old_mode=True)["common_hashes"] == 2


def test_fasta_counter_of_the_device_path_agrees_with_the_parser():
    """the device-side FASTA path counts records and total_bases on the host from the positions of the '>' bytes
    (FastaCounter); whatever the chunking, it must report what parse_fastx reports: random FASTA-shaped text with
    '>' inside lines, blank lines, CRLF, missing final newline, empty records, lines longer than a chunk"""
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGTACGTNacgt> \t\r-", dtype=np.uint8)
    for case in range(300):
        lines = []
        for i in range(int(rng.choice([1, 3, 20, 200]))):
            r = rng.random()
            if i == 0 or r < 0.15:
                lines.append(b">" + bytes(rng.choice(alpha, size=int(rng.integers(0, 30)))))
            elif r < 0.22:
                lines.append(b"")
            else:
                lines.append(bytes(rng.choice(alpha, size=int(rng.choice([1, 5, 60, 300])))))
        eol = [b"\n", b"\r\n"][int(rng.integers(0, 2))]
        data = eol.join(lines) + (eol if rng.random() < 0.6 else b"")
        want = H.fastx_scan(data)[:2]
        for chunk in (1, 7, 64, 4096, 1 << 20):
            assert H.fasta_count_chunked(data, chunk) == want, (case, chunk, data[:80])


def test_large_reads_split_over_threads_deliver_the_file(tmp_path):
    """FileSource::read splits requests of >= 16 MiB on a regular file over threads (pread); whatever the request
    size, thread count or file length, the bytes and their order are the file's, short tails and a rewind included"""
    rng = np.random.default_rng(5)
    M = 1 << 20
    for size in (0, 1, 2, 3 * M + 17, 16 * M, 16 * M + 2, 37 * M + 4099):
        data = rng.integers(0, 256, size=size, dtype=np.uint8).tobytes()
        p = tmp_path / ("f%d.bin" % size)
        p.write_bytes(data)
        for chunk, thr in ((64 * M, 4), (16 * M, 3), (17 * M + 5, 8), (1 * M, 4), (64 * M, 1)):
            got = H.read_file_probe(str(p), chunk, thr, size + 4096)
            assert got == data, (size, chunk, thr, len(got))


def _bgzf(data: bytes, block: int = 60000, eof_marker: bool = True) -> bytes:
    """what bgzip writes: gzip members of <= 64 KiB with their own size in the 'BC' extra subfield"""
    import struct
    import zlib
    out = []
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof_marker else [])
    for ch in chunks:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        c = co.compress(ch) + co.flush()
        hdr = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(c) + 25)
        out.append(hdr + c + struct.pack("<II", zlib.crc32(ch), len(ch)))
    return b"".join(out)


def test_bgzf_members_are_inflated_in_parallel_and_checked(monkeypatch):
    """BGZF input (bgzip): with more than one decompression thread the members are inflated concurrently; the records
    must be those of the plain text, a non-BGZF member hands over to the sequential reader, damage stays loud"""
    rng = np.random.default_rng(11)
    reads = [bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=int(rng.integers(1, 400)))) for _ in range(30000)]
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))
    want = H.fastx_scan(fq)
    assert want[0] == len(reads) and want[2] == 2
    z = _bgzf(fq)
    for thr in ("1", "2", "5"):
        F.debug_set(bgzf_threads=thr)
        assert H.fastx_scan(z) == want, thr
        assert H.fastx_scan(_bgzf(fq, 65536)) == want                      # full-size members
        assert H.fastx_scan(_bgzf(fq, 777, eof_marker=False)) == want      # thousands of tiny members, no EOF marker
        # a plain gzip member in the middle / at the start: the sequential reader takes over from there
        half = fq.index(b"@r15000\n")
        assert H.fastx_scan(_bgzf(fq[:half], eof_marker=False) + gzip.compress(fq[half:], 1)) == want
        assert H.fastx_scan(gzip.compress(fq[:half], 1) + _bgzf(fq[half:])) == want
        with pytest.raises(FinchError, match="empty input"):                 # only the EOF marker: empty, as for plain text
            H.fastx_scan(_bgzf(b""))
    F.debug_set(bgzf_threads="4")
    bad = bytearray(z)
    bad[len(z) // 2] ^= 0x55  # flips a bit inside some member's deflate data or trailer
    with pytest.raises(Exception):
        H.fastx_scan(bytes(bad))
    with pytest.raises(Exception):
        H.fastx_scan(z[:len(z) // 2])  # truncated inside a member


def test_truncated_gzip_is_an_error():
    """flate2 (needletail's decoder) reports UnexpectedEof for a gzip stream that ends inside a member; so must we --
    cut inside the deflate data, inside the trailer, and between two members (the only clean place)"""
    fq = b"".join(b"@r%d\nACGTACGTACGTTTGACCA\n+\nIIIIIIIIIIIIIIIIIII\n" % i for i in range(5000))
    z = gzip.compress(fq, 6)
    assert H.fastx_scan(z)[0] == 5000
    for cut in (len(z) // 2, len(z) - 1, len(z) - 5, len(z) - 9):
        with pytest.raises(FinchError):
            H.fastx_scan(z[:cut])
    two = z + gzip.compress(fq, 1)
    assert H.fastx_scan(two)[0] == 10000
    assert H.fastx_scan(two[:len(z)])[0] == 5000
    with pytest.raises(FinchError):
        H.fastx_scan(two[:len(z) + 40])


def test_truncated_bz2_and_xz_are_errors():
    import bz2
    import lzma
    fq = b"".join(b"@r%d\nACGTACGTACGTTTGACCA\n+\nIIIIIIIIIIIIIIIIIII\n" % i for i in range(20000))
    for comp in (bz2.compress(fq), lzma.compress(fq, preset=1)):
        assert H.fastx_scan(comp)[0] == 20000
        for cut in (len(comp) // 2, len(comp) - 3):
            with pytest.raises(FinchError):
                H.fastx_scan(comp[:cut])
    assert H.fastx_scan(bz2.compress(fq) + bz2.compress(fq))[0] == 40000  # concatenated streams, as needletail reads them


def test_decompressed_stream_is_the_text_whatever_the_request_size(monkeypatch):
    """gzip / BGZF / bzip2 / xz / plain: the bytes behind open_source equal the text for small requests (the parsers),
    requests of several MiB (device-side text paths; BGZF inflates straight into the caller's buffer) and odd sizes"""
    import bz2
    import lzma
    rng = np.random.default_rng(3)
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=97)), b"I" * 97)
                    for i in range(60000))  # ~13 MB
    images = {"plain": text, "gzip": gzip.compress(text, 1), "bgzf": _bgzf(text), "bgzf x3 batches": _bgzf(text, 9000), "bgzf+gzip": _bgzf(text[:5_000_000], eof_marker=False) + gzip.compress(text[5_000_000:], 1),
              "bz2": bz2.compress(text[:3_000_000]), "xz": lzma.compress(text[:3_000_000], preset=1)}
    for thr in ("1", "4"):
        F.debug_set(bgzf_threads=thr)
        for name, img in images.items():
            want = text if name in ("plain", "gzip", "bgzf", "bgzf x3 batches", "bgzf+gzip") else text[:3_000_000]
            for chunk in (4096, 1 << 20, (5 << 20) + 13, 64 << 20):
                assert H.source_probe(img, chunk, len(text) + 4096) == want, (name, thr, chunk)


def _naive_state_and_halo(text: bytes, off: int, k: int):
    """what a sketcher that starts at text[off] has to be told: (start_state, last k-1 kept bytes of the record so far)"""
    in_header, have, kept = False, False, bytearray()
    i = 0
    line_start = True
    while i < off:
        c = text[i]
        if line_start and c == 0x3E:
            in_header, have, kept = True, True, bytearray()
        if c == 0x0A:
            in_header = False
            line_start = True
        else:
            if not in_header and c not in b" \t\r":
                kept.append(c)
            line_start = False
        i += 1
    state = 2 if in_header else (0 if line_start else 1)
    return state, (bytes(kept[-(k - 1):]) if (k > 1 and have and not in_header) else b"")


def test_sharded_reader_chunks_tile_the_text_and_carry_the_right_halo():
    """finch_sketch_file_sharded's reader (host only): FASTA chunks are cut after newlines, tile the text, and each one
    carries the start state and the k-1 sequence bytes a sketcher needs to form the k-mers across the cut -- checked
    against a byte-by-byte walk of the text, with chunks small enough that a halo has to be chained through several of
    them; FASTQ chunks hold whole records.  Record count / total_bases agree with the parser's."""
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGTACGTACGTNacgt> \t\r", np.uint8)
    for case in range(40):
        lines = []
        for i in range(int(rng.choice([3, 40, 400]))):
            r = rng.random()
            if i == 0 or r < 0.1:
                lines.append(b">" + bytes(rng.choice(alpha, size=int(rng.integers(0, 30)))).replace(b"\r", b""))
            elif r < 0.2:
                lines.append(bytes(rng.choice(np.frombuffer(b" \t", np.uint8), size=int(rng.integers(0, 4)))))
            else:
                lines.append(bytes(rng.choice(alpha, size=int(rng.choice([1, 5, 30, 70, 200])))).replace(b"\r", b""))
        eol = [b"\n", b"\r\n"][case % 2]
        text = eol.join(lines) + (eol if case % 3 else b"")
        for k, chunk in ((21, 16), (31, 64), (5, 17), (32, 257), (21, 4096), (64, 100)):
            chunks, nrec, tb = H.shard_probe(text, k, chunk)
            pos = 0
            for off, ln, st, halo in chunks:
                assert off == pos and 0 < ln <= chunk
                pos += ln
                want_st, want_halo = _naive_state_and_halo(text, off, k)
                assert st == want_st, (case, k, chunk, off)
                if st == 0 and text[off:off + 1] == b">":
                    pass  # a header follows at once: whatever halo is passed is cut off by the record breaker
                else:
                    assert halo == want_halo, (case, k, chunk, off, halo, want_halo)
                if pos < len(text) and b"\n" in text[off:pos]:
                    assert text[pos - 1:pos] == b"\n"  # cut after a newline whenever the chunk holds one
            assert pos == len(text)
            n2, tb2, fmt = H.fastx_scan(text)
            assert (nrec, tb) == (n2, tb2) and fmt == 1
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"ACGT" * (1 + i % 9), b"@I+#" * (1 + i % 9)) for i in range(300))
    for chunk in (200, 1000, 1 << 20):
        chunks, _, _ = H.shard_probe(fq, 21, chunk)
        pos = 0
        for off, ln, st, halo in chunks:
            assert off == pos and halo == b""
            piece = fq[off:off + ln]
            assert piece.startswith(b"@r") and piece.endswith(b"\n") and piece.count(b"\n") % 4 == 0
            pos += ln
        assert pos == len(fq)


def _gzip_member(data: bytes, level=6, strategy=0, name=None, comment=None, extra=None, hcrc=False) -> bytes:
    """a gzip member with any of the optional header fields of RFC 1952"""
    import struct
    import zlib
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    body = co.compress(data) + co.flush()
    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    hdr = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\x00\xff"
    if extra is not None:
        hdr += struct.pack("<H", len(extra)) + extra
    if name is not None:
        hdr += name + b"\0"
    if comment is not None:
        hdr += comment + b"\0"
    if hcrc:
        hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    return hdr + body + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)


def test_the_library_s_own_inflate_decodes_what_zlib_writes(monkeypatch):
    """fh_inflate.h (the gzip path of the byte sources): stored / fixed / dynamic blocks, every compression level and
    strategy zlib has, data from incompressible to one long run, optional header fields, concatenated members, requests of
    1 byte to 16 MiB -- the delivered stream is the text.  Damaged input is an error: truncation at any point, flipped
    bits (caught by the decoder or by the member's CRC-32 / ISIZE), garbage after a member."""
    import zlib
    F.debug_set(bgzf_threads="1")  # the sequential reader (the parallel BGZF reader has its own test above)
    rng = np.random.default_rng(31)
    kinds = {
        "random": lambda n: rng.integers(0, 256, n, dtype=np.uint8).tobytes(),
        "dna": lambda n: rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n).tobytes(),
        "fastq": lambda n: (b"".join(b"@r%d\n%s\n+\n%s\n" % (i, rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=100).tobytes(),
                                                                 rng.choice(np.frombuffer(b"FF:,#", np.uint8), size=100).tobytes())
                                     for i in range(n // 210 + 1)))[:n],
        "runs": lambda n: (b"I" * 997 + b"\n" + b"AC" * 333) * (n // 1664 + 1),
        "zeros": lambda n: bytes(n),
    }
    n_checked = 0
    for kind, gen in kinds.items():
        for n in (0, 1, 5, 300, 70000, 700000):
            data = gen(n)[:n]
            for level, strategy in ((0, 0), (1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (6, zlib.Z_FILTERED)):
                if n > 70000 and (level, strategy) not in ((1, 0), (6, 0), (6, zlib.Z_FIXED)):
                    continue
                gz = _gzip_member(data, level, strategy)
                for chunk in (1 << 24, 4096, 1) if n <= 300 else (1 << 24, 65536 + 1):
                    assert H.source_probe(gz, chunk, n + 64) == data, (kind, n, level, strategy, chunk)
                    n_checked += 1
    assert n_checked > 300
    text = kinds["fastq"](200000)
    text = text[:text.rindex(b"\n@r") + 1]  # whole records
    # optional header fields, several members (bgzip-style and plain), an empty member in between
    multi = (_gzip_member(text[:50000], 6, 0, name=b"reads.fq", comment=b"made by a test", extra=b"XY\x03\x00abc", hcrc=True) +
             _gzip_member(b"") + _gzip_member(text[50000:120000], 1) + _gzip_member(text[120000:], 9, name=b"x"))
    for chunk in (1 << 24, 1000, 7):
        assert H.source_probe(multi, chunk, len(text) + 64) == text
    assert H.fastx_scan(multi)[2] == 2
    # damage
    gz = _gzip_member(text[:30000], 6)
    for cut in list(range(2, 40)) + list(range(40, len(gz) - 1, 97)) + [len(gz) - 9, len(gz) - 5, len(gz) - 1]:
        with pytest.raises(FinchError, match="corrupt"):  # (below two bytes there is no gzip magic to sniff)
            H.source_probe(gz[:cut], 1 << 20, 40000)
    caught = 0
    for trial in range(400):
        b = bytearray(gz)
        i = int(rng.integers(10, len(b)))
        b[i] ^= 1 << int(rng.integers(0, 8))
        try:
            got = H.source_probe(bytes(b), 1 << 20, 40000)
            assert False, "a flipped bit at %d went unnoticed (%d bytes delivered)" % (i, len(got))
        except FinchError:
            caught += 1
    assert caught == 400
    for tail in (b"\0", b"garbage", b"\x1f\x8b\x08"):
        with pytest.raises(FinchError, match="corrupt"):
            H.source_probe(gz + tail, 1 << 20, 40000)
    with pytest.raises(FinchError, match="corrupt"):
        H.source_probe(gz[:3] + b"\xe0" + gz[4:], 1 << 20, 40000)  # reserved FLG bits
    # and the zlib-based reader (option zlib_inflate=1) is still there for A/B runs: same bytes (checked in a child process:
    # the switch is read once per process)
    import subprocess, sys, os
    code = ("import sys; sys.path.insert(0, %r); from finch_rs_amd import host as H; d = open(sys.argv[1], 'rb').read(); "
            "sys.stdout.buffer.write(H.source_probe(d, 100000, 1 << 20))" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".gz") as f:
        f.write(multi)
        f.flush()
        env = F.debug_env(zlib_inflate="1", bgzf_threads="1")
        assert subprocess.run([sys.executable, "-c", code, f.name], env=env, stdout=subprocess.PIPE, check=True).stdout == text


def test_bgzf_reader_for_the_device_side_inflate():
    """the member tables fh_push_bgzf_fastq is fed (BgzfSource::raw_batch): every member inflates, where the table says, to
    the text -- whatever limit ends a batch (buffer, member count, text budget), for tiny and full-size members, with and
    without the EOF marker; damage and foreign members are refused"""
    rng = np.random.default_rng(21)
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=120)),
                                              bytes(rng.integers(35, 74, size=120, dtype=np.uint8))) for i in range(12000))  # ~3 MB
    for block, eof_marker in ((60000, True), (65280, False), (1500, True)):
        z = _bgzf(text, block, eof_marker)
        for buf_bytes, max_members, budget in ((1 << 20, 4096, 1 << 30), (300_000, 3, 1 << 30), (1 << 20, 4096, 200_000), (4 << 20, 16384, 70_000)):
            got, batches, fb = H.bgzf_batch_probe(z, buf_bytes, max_members, budget, len(text) + 65536)
            assert got == text and fb == ord("@"), (block, buf_bytes, max_members, budget)
            assert batches > 1
    z = _bgzf(text, 60000)
    with pytest.raises(FinchError):  # a plain gzip member behind BGZF ones
        H.bgzf_batch_probe(z + gzip.compress(b"@x\nA\n+\nI\n"), 1 << 20, 4096, 1 << 30, len(text) + 65536)
    with pytest.raises(FinchError):  # truncated
        H.bgzf_batch_probe(z[:len(z) // 2], 1 << 20, 4096, 1 << 30, len(text) + 65536)
    with pytest.raises(FinchError):  # stray bytes at the end
        H.bgzf_batch_probe(z + b"\x1f\x8b\x08", 1 << 20, 4096, 1 << 30, len(text) + 65536)
    dmg = bytearray(z)
    dmg[len(dmg) // 3] ^= 0x40
    with pytest.raises(FinchError):
        H.bgzf_batch_probe(bytes(dmg), 1 << 20, 4096, 1 << 30, len(text) + 65536)


def test_the_reader_of_the_device_side_gzip_inflate_skips_the_header_and_hands_the_stream_over_in_pieces():
    """finch_gzip_probe (no device): the probe that decides whether a file goes to fh_push_gzip_fastq -- first byte of the text,
    length of the RFC 1952 header with any of its optional fields -- and the bytes behind the header exactly as the file has
    them, whatever the piece size; BGZF, other formats and text that cannot be reached say -1"""
    import struct
    import zlib
    rng = np.random.default_rng(5)
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=100)), b"I" * 100) for i in range(3000))
    body = zlib.compress(text, 6)[2:-4] + struct.pack("<II", zlib.crc32(text), len(text))

    def image(flg, extra=b"\x07\x00extra!!"[:9], name=b"reads.fastq\0", comment=b"made by a test\0"):
        h = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\x00\x03"
        if flg & 4:
            h += struct.pack("<H", len(extra)) + extra
        if flg & 8:
            h += name
        if flg & 16:
            h += comment
        if flg & 2:
            h += struct.pack("<H", zlib.crc32(h) & 0xFFFF)
        return h, h + body
    for flg in (0, 2, 4, 8, 16, 8 | 16, 2 | 4 | 8 | 16):
        h, img = image(flg)
        for piece in (1, 4097, 1 << 20):
            if piece == 1 and flg not in (0, 30):
                continue
            hdr_len, first, n, crc = H.gzip_probe(img, piece)
            assert (hdr_len, first, n, crc) == (len(h), ord("@"), len(body), zlib.crc32(body)), (flg, piece)
    assert H.gzip_probe(gzip.compress(b">g\nACGT\n" * 5000, 6))[1] == ord(">")
    assert H.gzip_probe(_bgzf(text, 60000))[1] == -1                      # BGZF: the other path's
    assert H.gzip_probe(b"@r\nACGT\n+\nIIII\n" * 100)[1] == -1        # not compressed at all
    assert H.gzip_probe(image(0)[1][:30])[1] == -1                        # too short to tell
    assert H.gzip_probe(image(8, name=b"x" * 70000)[1])[1] == -1          # a name that never ends within the probe's reach
    assert H.gzip_probe(b"\x1f\x8b\x08\x00" + b"\0" * 6 + b"\xff" * 5000)[1] == -1  # no DEFLATE stream behind the header
    assert H.gzip_probe(b"\x1f\x8b\x08\xe0" + b"\0" * 6 + body)[1] == -1  # reserved flag bits


def test_one_gzip_member_decoded_by_several_threads(monkeypatch):
    """plain gzip with read threads to spare (fh_pargz.h): block starts found by search, chunks decoded with markers for the
    unknown window, stitched and checked against the member's CRC-32 -- the text must be zlib's for every chunk size,
    compression level and request size, for several members, stored blocks, binary data (no block is ever found: one
    thread decodes), and damage must stay loud"""
    import zlib
    rng = np.random.default_rng(33)
    g = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=100_000)
    recs = []
    for i in range(30000):
        st = int(rng.integers(0, len(g) - 150))
        recs.append(b"@read%d\n%s\n+\n%s\n" % (i, g[st:st + 150].tobytes(), bytes(rng.integers(35, 74, size=150, dtype=np.uint8))))
    text = b"".join(recs)  # ~9.5 MB
    unique = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=3_000_000))  # markers fade quickly in this one
    fasta = b">chr1 test\n" + b"\n".join(unique[i:i + 70] for i in range(0, len(unique), 70)) + b"\n"
    F.debug_set(bgzf_threads="4")
    for chunk in ("40000", "250000", "4194304"):
        F.debug_set(pargz_chunk=chunk)
        for level in (1, 6, 9):
            z = gzip.compress(text, level)
            for req in (4096, (5 << 20) + 13, 64 << 20):
                assert H.source_probe(z, req, len(text) + 4096) == text, (chunk, level, req)
        assert H.source_probe(gzip.compress(fasta, 6), 1 << 20, len(fasta) + 4096) == fasta
        # members behind the first go through the sequential reader; an empty one in between is fine
        multi = gzip.compress(text[:4_000_000], 6) + gzip.compress(b"") + gzip.compress(text[4_000_000:], 1)
        assert H.source_probe(multi, 1 << 20, len(text) + 4096) == text
        # stored blocks only, the fixed code only, and a mix of both with dynamic blocks
        for kw in (dict(level=0), dict(level=6, strategy=zlib.Z_FIXED)):
            co = zlib.compressobj(kw.get("level", 6), zlib.DEFLATED, 31, 8, kw.get("strategy", zlib.Z_DEFAULT_STRATEGY))
            z = co.compress(text[:3_000_000]) + co.flush()
            assert H.source_probe(z, 1 << 20, len(text)) == text[:3_000_000], kw
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        z = co.compress(text[:2_000_000]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(text[2_000_000:5_000_000]) + co.flush(zlib.Z_SYNC_FLUSH) + \
            co.compress(text[5_000_000:]) + co.flush()
        assert H.source_probe(z, 1 << 20, len(text) + 4096) == text
        # not text: the search finds no block it believes in, the first thread decodes everything
        blob = bytes(rng.integers(0, 256, size=1_500_000, dtype=np.uint8)) + bytes(2_000_000) + text[:1_000_000]
        assert H.source_probe(gzip.compress(blob, 6), 1 << 20, len(blob) + 4096) == blob
        # damage: a flipped bit somewhere in the middle, a cut, a wrong checksum
        z = bytearray(gzip.compress(text, 6))
        for cut in (len(z) // 2, len(z) - 5):
            with pytest.raises(FinchError):
                H.source_probe(bytes(z[:cut]), 1 << 20, len(text) + 4096)
        bad = bytearray(z)
        bad[-6] ^= 1
        with pytest.raises(FinchError):
            H.source_probe(bytes(bad), 1 << 20, len(text) + 4096)
        for where in (len(z) // 3, (2 * len(z)) // 3):
            bad = bytearray(z)
            bad[where] ^= 0x20
            with pytest.raises(FinchError):
                H.source_probe(bytes(bad), 1 << 20, len(text) + 4096)
    # tiny inputs
    for t in (b"", b"@r\nA\n+\nI\n", text[:70000]):
        assert H.source_probe(gzip.compress(t), 4096, len(t) + 64) == t
    # the records are what the host parser makes of the plain text
    F.debug_set(pargz_chunk="100000")
    assert H.fastx_scan(gzip.compress(text, 6)) == H.fastx_scan(text)


def test_gzip_header_fields_in_front_of_a_member_decoded_in_parallel(monkeypatch):
    """RFC 1952 header variants (FNAME as `gzip file` writes it, FEXTRA that is not BGZF's, FCOMMENT, FHCRC) in front of a
    member the parallel reader takes"""
    import struct
    import zlib
    F.debug_set(bgzf_threads="4")
    F.debug_set(pargz_chunk="60000")
    rng = np.random.default_rng(1)
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=100)),
                                              bytes(rng.integers(35, 74, size=100, dtype=np.uint8))) for i in range(20000))
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(text) + co.flush()
    trailer = struct.pack("<II", zlib.crc32(text), len(text))

    def hdr(flg, extra=b"", name=b"", comment=b""):
        h = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\x00\x03"
        if flg & 4:
            h += struct.pack("<H", len(extra)) + extra
        if flg & 8:
            h += name + b"\0"
        if flg & 16:
            h += comment + b"\0"
        if flg & 2:
            h += struct.pack("<H", zlib.crc32(h) & 0xFFFF)
        return h

    for h in (hdr(0), hdr(8, name=b"reads.fastq"), hdr(4, extra=b"XY\x03\0abc"),
              hdr(4 | 8 | 16 | 2, extra=b"AB\x02\0zz", name=b"x" * 300, comment=b"made by a test")):
        z = h + body + trailer
        assert zlib.decompress(z, 31) == text
        assert H.source_probe(z, 1 << 20, len(text) + 4096) == text


def _tiny_dynamic_block(extra_hclen):
    """raw DEFLATE: ONE final dynamic-Huffman block holding the literal 'A', written so that the header's last code-length
    code sits within a few bits of the end of the stream (1-bit literal, 1-bit end-of-block, one distance code of length 0).
    extra_hclen more (zero) entries of the code-length-code table move the bit alignment of everything behind them."""
    bits = []

    def put(v, n):  # LSB first
        for i in range(n):
            bits.append((v >> i) & 1)

    def code(c, n):  # Huffman codes go MSB first
        for i in range(n - 1, -1, -1):
            bits.append((c >> i) & 1)
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    cl = {18: 1, 0: 2, 1: 2}  # code-length alphabet: 18 -> '0', 0 -> '10', 1 -> '11'
    hclen = 18 + min(extra_hclen, 1)  # symbol 1 is entry 17 of `order`: at least 18 entries
    put(1, 1); put(2, 2)  # BFINAL, BTYPE = dynamic
    put(0, 5); put(0, 5); put(hclen - 4, 4)  # HLIT = 257, HDIST = 1
    for s in order[:hclen]:
        put(cl.get(s, 0), 3)
    canon = {18: (0, 1), 0: (2, 2), 1: (3, 2)}
    def z(n):  # n zeros, 11..138
        code(*canon[18]); put(n - 11, 7)
    z(65); code(*canon[1]); z(138); z(52); code(*canon[1])  # lengths of literals 0..256: 'A' and end-of-block get 1 bit
    code(*canon[0])                                           # the one distance code: length 0
    code(0, 1); code(1, 1)                                    # 'A', end of block
    while len(bits) % 8:
        bits.append(0)
    return bytes(sum(b << i for i, b in enumerate(bits[j:j + 8])) for j in range(0, len(bits), 8))


def test_dynamic_block_that_ends_within_bits_of_its_header(monkeypatch):
    """fh_inflate.h used to ask for more input when fewer than 7 bits were left in front of a code-length code -- before
    looking at the code -- and rejected members zlib accepts: a BGZF member's input ends exactly where its DEFLATE stream does"""
    import struct
    import zlib
    F.debug_set(bgzf_threads="4")
    for extra in (0, 1):
        raw = _tiny_dynamic_block(extra)
        d = zlib.decompressobj(-15)
        assert d.decompress(raw) == b"A" and d.eof
        # several members: a FASTQ record split over members, the tiny dynamic block being one of them
        parts = [b"@r\nAC", b"A", b"GT\n+\nIIII\n"]
        bg = b""
        for i, t in enumerate(parts):
            if i == 1:
                body = raw
            else:
                c = zlib.compressobj(6, zlib.DEFLATED, -15)
                body = c.compress(t) + c.flush()
            bg += (b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(body) + 25) + body +
                   struct.pack("<II", zlib.crc32(t), len(t)))
        bg += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")  # BGZF end-of-file marker
        assert zlib.decompress(bg[:len(bg) - 28], 31) == parts[0]  # (framing sanity: the first member is a gzip member)
        assert H.source_probe(bg, 1 << 16, 4096) == b"".join(parts)


def test_parallel_gzip_keeps_going_through_a_very_compressible_member(tmp_path):
    """the guard against false starts (a chunk that inflates past 64 MiB and 256x its compressed size) must not fire on a
    chunk whose start is known: 70 MB of N inflate from ~70 KB, and the parallel reader must neither hand the stream to the
    sequential one (everything decoded twice) nor -- on a pipe -- call it corrupt"""
    import os
    import subprocess
    import sys
    text = b">n\n" + b"N" * (70 << 20) + b"\n"
    z = gzip.compress(text, 6)
    p = tmp_path / "n.fa.gz"
    p.write_bytes(z)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys, hashlib; sys.path.insert(0, %r)\nfrom finch_rs_amd import host as H\n"
            "d = H.source_probe(open(%r, 'rb').read(), 1 << 22, %d)\nprint(len(d), hashlib.md5(d).hexdigest())" % (root, str(p), len(text) + 64))
    r = subprocess.run([sys.executable, "-c", prog], env=F.debug_env(trace="1", bgzf_threads="4"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    import hashlib
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == [str(len(text)), hashlib.md5(text).hexdigest()]
    assert "parallel gzip" in r.stderr and "sequential reader took over" not in r.stderr, r.stderr[-2000:]


@pytest.mark.parametrize("k", [15, 21, 44, 45, 64])
def test_kmer_bytes_of_every_length_survive_the_three_formats(k):
    """a k-mer's bytes live inline in its record up to 44 bytes and on the heap beyond (fh_host_model.h KmerBytes): both
    kinds through finch_sketches_from_arrays, the .sk / .bsk / .msh writers and their readers, and the filters' reordering"""
    rng = np.random.default_rng(k)
    n = 300
    counts = (rng.poisson(3, n) + 1).astype(np.uint32)
    kc = kc_of(counts, (counts * rng.random(n)).astype(np.uint32))
    kc["hash"] = np.sort(rng.integers(1, 2**62, n).astype(np.uint64))
    km = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, k))
    params = SketchParams.mash(n, n, True, k, 0)
    sk = H.sketches_from_arrays("long", 10, 20, kc, km, params, H.FilterParams(False))
    got = sk.sketch(0)
    assert np.array_equal(got.arrays[0], kc) and np.array_equal(got.arrays[1], km)
    for back in (H.sketches_from_json(sk.to_json()), H.sketches_from_bsk(sk.to_bsk())):
        g2 = back.sketch(0)
        assert np.array_equal(g2.arrays[0]["hash"], kc["hash"]) and np.array_equal(g2.arrays[0]["count"], kc["count"])
        assert np.array_equal(g2.arrays[1], km)
    m = H.sketches_from_msh(sk.to_msh()).sketch(0)  # (.msh carries no k-mer bytes)
    assert np.array_equal(m.arrays[0]["hash"], kc["hash"])
    fp = sk.apply_filters(0, H.FilterParams(True, (2, None), 0.0, 0.0))
    keep = kc["count"] >= 2
    got = sk.sketch(0)
    assert np.array_equal(got.arrays[0], kc[keep]) and np.array_equal(got.arrays[1], km[keep]) and fp.filter_on
