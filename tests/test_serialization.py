"""The sketch file formats next to `.sk`: `.bsk` (finch.capnp) and `.msh` (mash.capnp) writers and readers, the `.sk`
reader and open_sketch_file -- lib/src/serialization/{mod,mash,json}.rs and lib.rs:96-118 of the reference.

There is no Cap'n Proto runtime in the image, so the hand-written encoders are pinned three ways:
  * byte level: the words of a small message are compared with the layout the schemas' field ordinals give, slot by slot as
    capnpc computed them for the reference (finch_capnp.rs / mash_capnp.rs line numbers in tests/capnp_mini.py);
  * an independent decoder (tests/capnp_mini.py: spec-driven, knows nothing of the product) reads back the CLI-golden
    sketch (cli/tests/test_cli.rs:99-143) and a 2 M-hash sketch, field for field;
  * the product's own readers round-trip the same files -- also after the messages were re-laid over several segments with
    far and double-far pointers, which is how the reference's own writer lays large sketches out.
CPU only: the sketches come from the oracle / from arrays."""
import json
import os
import struct

import numpy as np
import pytest

import capnp_mini as CM
from finch_rs_amd import host as H
from finch_rs_amd.sketch_schemes import FinchError, KC_DTYPE, SketchParams
from oracle import oracle as O


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as G
    G.build()


def golden_sketches(golden_dir, params=None, filters=None, name=None):
    """the sketch of cli/tests/data/query.fa (k=21, n=10, seed 0) as the oracle computes it, wrapped as the host layer's Vec<Sketch>"""
    path = os.path.join(golden_dir, "query.fa")
    data = open(path, "rb").read()
    o = O.OracleSketcher(O.MASH, 10, 21, 0)
    assert o.sketch_stream(data) == 1
    kc, km = o.to_vec()
    tb, tk = o.total_bases_and_kmers()
    params = params or SketchParams.mash(2000, 10, False, 21, 0)
    return H.sketches_from_arrays(name or path, tb, tk, kc, km, params, filters or H.FilterParams(False)), kc, km, (tb, tk)


def test_bsk_layout_word_by_word(golden_dir):
    """write_finch_file (serialization/mod.rs:123-166) on a sketch with distinctive values: every word of the message is
    where finch.capnp's field ordinals put it"""
    kc = np.zeros(2, dtype=KC_DTYPE)
    kc["hash"], kc["count"], kc["extra_count"] = [0x1111111111111111, 0xFFFFFFFFFFFFFFFE], [7, 0xAABBCCDD], [3, 0x11223344]
    km = np.frombuffer(b"ACGTA" + b"TTTTG", np.uint8).reshape(2, 5)
    p = SketchParams.mash(0x0102030405060708, 0x1112131415161718, True, 5, 0xA1A2A3A4A5A6A7A8)
    f = H.FilterParams(True, (9, 0x55667788), 0.25, 0.125)
    sk = H.sketches_from_arrays("nm", 0x2122232425262728, 0x3132333435363738, kc, km, p, f)
    sk.set_comment(0, "cmt")
    raw = sk.to_bsk()
    assert struct.unpack_from("<II", raw, 0) == (0, (len(raw) - 8) // 8) and len(raw) % 8 == 0  # one segment, framing of write_message
    w = np.frombuffer(raw, dtype="<u8", offset=8)

    def sptr(off, dw, pw):
        return (off << 2) | (dw << 32) | (pw << 48)

    def lptr(off, elem, cnt):
        return (off << 2) | 1 | (elem << 32) | (cnt << 35)
    assert w[0] == sptr(0, 0, 1)                    # root -> Multisketch {0 data, 1 pointer}
    assert w[1] == lptr(0, 7, 7)                    # sketches: composite list, 1 x 7 words
    assert w[2] == (1 << 2) | (2 << 32) | (5 << 48)  # tag: 1 element of Sketch {2, 5}
    S = 3                                           # the Sketch struct
    assert w[S] == 0x2122232425262728 and w[S + 1] == 0x3132333435363738  # seqLength @1, numValidKmers @2
    nxt = S + 7
    assert w[S + 2] == lptr(nxt - (S + 2) - 1, 2, 3) and w[nxt].tobytes()[:3] == b"nm\0"     # name @0: Text, NUL counted
    nxt += 1
    assert w[S + 3] == lptr(nxt - (S + 3) - 1, 2, 4) and w[nxt].tobytes()[:4] == b"cmt\0"    # comment @3
    nxt += 1
    assert w[S + 4] == lptr(nxt - (S + 4) - 1, 7, 8)                                         # hashes @4: 2 x KmerCount {2, 2}
    assert w[nxt] == (2 << 2) | (2 << 32) | (2 << 48)
    K = nxt + 1
    for j in range(2):
        e = K + 4 * j
        assert w[e] == kc["hash"][j]                                                         # hash @0
        assert w[e + 1] == int(kc["count"][j]) | (int(kc["extra_count"][j]) << 32)           # count @2, extraCount @3
        assert w[e + 3] == 0                                                                 # label @4: None -> null
    d0 = K + 8
    assert w[K + 2] == lptr(d0 - (K + 2) - 1, 2, 5) and w[d0].tobytes()[:5] == b"ACGTA"       # kmer @1: Data
    assert w[K + 6] == lptr(d0 + 1 - (K + 6) - 1, 2, 5) and w[d0 + 1].tobytes()[:5] == b"TTTTG"
    F = d0 + 2                                                                               # FilterParams {4, 0}
    assert w[S + 5] == sptr(F - (S + 5) - 1, 4, 0)
    assert w[F] == 1 | (9 << 32)                      # filtered @0 = bit 0, lowAbunFilter @1 = u32 slot 1
    assert w[F + 1] == 0x55667788                     # highAbunFilter @2 = u32 slot 2
    assert w[F + 2] == struct.unpack("<Q", struct.pack("<d", 0.25))[0] and w[F + 3] == struct.unpack("<Q", struct.pack("<d", 0.125))[0]
    P = F + 4                                                                                # SketchParams {5, 0}
    assert w[S + 6] == sptr(P - (S + 6) - 1, 5, 0)
    assert w[P] == 0 | (5 << 16) | (1 << 24)          # sketchMethod murmurHash3 = 0 (u16 slot 0), kmerLength u8 slot 2, noStrict bit 24
    assert (w[P + 1], w[P + 2], w[P + 3], w[P + 4]) == (0x0102030405060708, 0xA1A2A3A4A5A6A7A8, 0x1112131415161718, 0)
    assert len(w) == P + 5
    # scaled: method 1, scale in f64 slot 4, finalSize / noStrict untouched (set_sketch_params, mod.rs:76-86)
    sk2 = H.sketches_from_arrays("s", 1, 2, kc, km, SketchParams.scaled(77, 5, 0.001, 3), H.FilterParams(False))
    w2 = np.frombuffer(sk2.to_bsk(), dtype="<u8", offset=8)
    assert w2[-5] == 1 | (5 << 16) and (w2[-4], w2[-3], w2[-2]) == (77, 3, 0)
    assert w2[-1] == struct.unpack("<Q", struct.pack("<d", 0.001))[0]
    # filters off / unbounded: filtered false, lowAbunFilter 0, highAbunFilter u32::MAX (mod.rs:150-153)
    assert w2[-9] == 0 and w2[-8] == 0xFFFFFFFF


def test_msh_layout_word_by_word():
    """write_mash_file (serialization/mash.rs:12-58) against mash.capnp's ordinals"""
    kc = np.zeros(3, dtype=KC_DTYPE)
    kc["hash"], kc["count"], kc["extra_count"] = [5, 6, 0xFFFFFFFFFFFFFFFF], [1, 2, 0xFFFFFFFF], [0, 1, 2]
    km = np.frombuffer(b"ACGTA" * 3, np.uint8).reshape(3, 5)
    sk = H.sketches_from_arrays("ref1", 1000, 900, kc, km, SketchParams.mash(3, 3, False, 5, 0x1_0000_002B), H.FilterParams(False))
    sk.append(H.sketches_from_arrays("r2", 5, 4, kc[:1], km[:1], SketchParams.mash(3, 3, False, 5, 0x1_0000_002B), H.FilterParams(False)))
    raw = sk.to_msh()
    w = np.frombuffer(raw, dtype="<u8", offset=8)
    assert struct.unpack_from("<II", raw, 0) == (0, len(w))
    assert w[0] == (3 << 32) | (4 << 48)              # root -> MinHash {3, 4}
    M = 1
    assert w[M] == 5 | (5 << 32)                      # kmerSize @0, windowSize @1 = k
    assert w[M + 1] == 3 | (1 << 32)                  # minHashesPerWindow @2 = largest sketch; concatenated @3 = bit 96 set; 97, 98 clear
    assert w[M + 2] == (0x2B ^ 42) << 32              # error @6 = 0.0f (slot 4); hashSeed @10 = (seed as u32) XOR its default 42 (slot 5)
    assert w[M + 3] == 0 and w[M + 4] == 0            # referenceListOld @4, locusList @5: null
    A = M + 7
    assert w[M + 5] == ((A - (M + 5) - 1) << 2) | 1 | (2 << 32) | (5 << 35) and w[A].tobytes()[:5] == b"ACGT\0"   # alphabet @8
    R = A + 1
    assert w[M + 6] == ((R - (M + 6) - 1) << 2) | (0 << 32) | (1 << 48)      # referenceList @11 -> ReferenceList {0, 1}
    assert w[R] == (0 << 2) | 1 | (7 << 32) | (20 << 35)                    # references: composite, 2 x Reference {3, 7}
    assert w[R + 1] == (2 << 2) | (3 << 32) | (7 << 48)
    E = R + 2
    assert (w[E], w[E + 1], w[E + 2]) == (0, 1000, 900)                     # length @2 unset, length64 @7, numValidKmers @9
    assert w[E + 3] == 0 and w[E + 4] == 0 and w[E + 7] == 0                # sequence, quality, hashes32: null
    m = CM.Message(raw).root(CM.MASH, "MinHash")
    refs = m["referenceList"]["references"]
    assert [r["name"] for r in refs] == [b"ref1", b"r2"] and refs[0]["comment"] == b""
    assert refs[0]["hashes64"].tolist() == [5, 6, 0xFFFFFFFFFFFFFFFF] and refs[0]["counts32"].tolist() == [1, 2, 0xFFFFFFFF]
    assert refs[1]["hashes64"].tolist() == [5] and refs[1]["length64"] == 5 and refs[1]["numValidKmers"] == 4
    assert (m["kmerSize"], m["windowSize"], m["minHashesPerWindow"], m["hashSeed"], m["alphabet"]) == (5, 5, 3, 0x2B, b"ACGT")
    assert m["concatenated"] and not m["noncanonical"] and not m["preserveCase"] and m["error"] == 0.0
    assert not m["referenceListOld"]["_present"]


def check_bsk_against(msg_sketch, name, totals, kc, km, params: SketchParams, filt: H.FilterParams, comment=b""):
    assert msg_sketch["name"] == name.encode() and msg_sketch["comment"] == comment
    assert (msg_sketch["seqLength"], msg_sketch["numValidKmers"]) == totals
    h = msg_sketch["hashes"]
    assert np.array_equal(h["hash"], kc["hash"]) and np.array_equal(h["count"], kc["count"]) and np.array_equal(h["extraCount"], kc["extra_count"])
    assert np.array_equal(np.asarray(h["kmer"]), km) and not h["has_label"].any()
    sp = msg_sketch["sketchParams"]
    if params.kind == "mash":
        assert (sp["sketchMethod"], sp["kmerLength"], sp["kmersToSketch"], sp["hashSeed"], sp["finalSize"], sp["noStrict"], sp["scale"]) == \
            (0, params.kmer_length, params.kmers_to_sketch, params.hash_seed, params.final_size, params.no_strict, 0.0)
    else:
        assert (sp["sketchMethod"], sp["kmerLength"], sp["kmersToSketch"], sp["hashSeed"], sp["finalSize"], sp["noStrict"], sp["scale"]) == \
            (1, params.kmer_length, params.kmers_to_sketch, params.hash_seed, 0, False, params.scale)
    fp = msg_sketch["filterParams"]
    lo, hi = filt.abun_filter
    assert (fp["filtered"], fp["lowAbunFilter"], fp["highAbunFilter"], fp["errFilter"], fp["strandFilter"]) == \
        (bool(filt.filter_on), lo or 0, hi if hi is not None else 0xFFFFFFFF, filt.err_filter, filt.strand_filter)


def same_sketch(a: H.Sketch, b: H.Sketch, kmers=True, extra=True):
    assert (a.name, a.seq_length, a.num_valid_kmers, a.comment) == (b.name, b.seq_length, b.num_valid_kmers, b.comment)
    assert np.array_equal(a.arrays[0]["hash"], b.arrays[0]["hash"]) and np.array_equal(a.arrays[0]["count"], b.arrays[0]["count"])
    if extra:
        assert np.array_equal(a.arrays[0]["extra_count"], b.arrays[0]["extra_count"])
    if kmers:
        assert np.array_equal(a.arrays[1], b.arrays[1])


def test_cli_golden_sketch_through_every_format(golden_dir, tmp_path):
    """cli/tests/test_cli.rs:20-149 writes query.fa's sketch as .sk, .bsk and .msh and reads them back; here the files are
    written by the product, decoded by the independent reader, and read back by the product's readers"""
    vec = json.load(open(os.path.join(golden_dir, "reference_vectors.json")))["test_cli_rs_99_143"]
    params = SketchParams.mash(2000, 10, False, 21, 0)
    filt = H.FilterParams(False, (None, None), 0.21, 1.0)  # what `finch sketch` passes for FASTA input (filtering stays off)
    sk, kc, km, totals = golden_sketches(golden_dir, params, filt)
    assert [bytes(r).decode() for r in km] == vec["kmers"]
    name = os.path.join(golden_dir, "query.fa")
    # .bsk
    raw = sk.to_bsk()
    m = CM.Message(raw)
    assert m.total_bytes == len(raw)
    root = m.root(CM.FINCH, "Multisketch")
    assert len(root["sketches"]) == 1
    check_bsk_against(root["sketches"][0], name, totals, kc, km, params, filt)
    back = H.sketches_from_bsk(raw)
    assert len(back) == 1 and back.params_of(0) == params
    same_sketch(back.sketch(0), sk.sketch(0))
    assert back.sketch(0).filter_params == H.FilterParams(False, (None, None), 0.21, 1.0)
    assert [h.kmer.decode() for h in back.sketch(0).hashes] == vec["kmers"]   # test_cli.rs:99-108 through the binary format
    # .msh (test_cli.rs:111-143: hashes and counts survive, k-mers do not)
    rawm = sk.to_msh()
    mm = CM.Message(rawm).root(CM.MASH, "MinHash")
    ref = mm["referenceList"]["references"][0]
    assert np.array_equal(ref["hashes64"], kc["hash"]) and np.array_equal(ref["counts32"], kc["count"])
    assert (ref["name"], ref["length64"], ref["numValidKmers"]) == (name.encode(), totals[0], totals[1])
    assert (mm["kmerSize"], mm["hashSeed"], mm["minHashesPerWindow"]) == (21, 0, 10)
    backm = H.sketches_from_msh(rawm)
    pm = backm.params_of(0)  # mash.rs:66-74
    assert (pm.kind, pm.kmers_to_sketch, pm.final_size, pm.no_strict, pm.kmer_length, pm.hash_seed) == ("mash", 0, 0, True, 21, 0)
    bm = backm.sketch(0)
    same_sketch(bm, sk.sketch(0), kmers=False, extra=False)
    assert np.array_equal(bm.arrays[0]["extra_count"], kc["count"] // 2)  # mash.rs:116
    assert all(h.kmer == b"" for h in bm.hashes) and bm.filter_params == H.FilterParams(False)
    # .sk
    js = sk.to_json()
    backj = H.sketches_from_json(js)
    pj = backj.params_of(0)  # json.rs:169-175: sketchSize = expected_size() = final_size for both fields, no_strict true
    assert (pj.kind, pj.kmers_to_sketch, pj.final_size, pj.no_strict, pj.kmer_length, pj.hash_seed) == ("mash", 10, 10, True, 21, 0)
    bj = backj.sketch(0)
    same_sketch(bj, sk.sketch(0), extra=False)
    assert np.array_equal(bj.arrays[0]["extra_count"], kc["count"] // 2)  # json.rs:124
    assert bj.filter_params == H.FilterParams(False)  # filter_on false -> empty map -> Some(false), filters at 0 (filtering.rs:110-134)
    # files, by name (lib.rs:96-118, main.rs:53-70)
    for ext, ref_sk in ((".bsk", back), (".msh", backm), (".sk", backj), (".json", backj)):
        p = str(tmp_path / ("out" + ext))
        sk.write(p)
        got = H.open_sketch_file(p)
        assert got.params_of(0) == ref_sk.params_of(0)
        same_sketch(got.sketch(0), ref_sk.sketch(0))
    assert open(str(tmp_path / "out.bsk"), "rb").read() == raw and open(str(tmp_path / "out.sk")).read() == js
    with pytest.raises(FinchError, match=r"File suffix is not \*\.bsk, \*\.msh, or \*\.sk"):
        sk.write(str(tmp_path / "out.txt"))
    (tmp_path / "x.txt").write_bytes(raw)
    with pytest.raises(FinchError, match=r"File suffix is not \*\.bsk, \*\.msh, or \*\.sk"):
        H.open_sketch_file(str(tmp_path / "x.txt"))
    with pytest.raises(FinchError, match="Error opening"):
        H.open_sketch_file(str(tmp_path / "missing.bsk"))
    (tmp_path / "broken.sk").write_text(js[:-20])
    with pytest.raises(FinchError, match="Error parsing"):
        H.open_sketch_file(str(tmp_path / "broken.sk"))


def big_sketch(n=2_000_000, k=31, seed=5):
    rng = np.random.default_rng(seed)
    kc = np.zeros(n, dtype=KC_DTYPE)
    kc["hash"] = np.sort(rng.integers(0, 2**63, n, dtype=np.uint64) * 2 + rng.integers(0, 2, n, dtype=np.uint64))
    kc["count"] = rng.integers(1, 2**32, n, dtype=np.uint64).astype(np.uint32)
    kc["extra_count"] = (kc["count"] * rng.random(n)).astype(np.uint32)
    km = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, k))
    return kc, km


def test_two_million_hash_sketch_round_trips(tmp_path):
    """BASELINE configs[2]'s oversketch (k=31, 2 000 000 hashes, ~128 MB as .bsk) through both binary formats"""
    kc, km = big_sketch()
    params = SketchParams.mash(2_000_000, 10_000, False, 31, 0)
    filt = H.FilterParams(True, (3, None), 0.31, 0.1)
    sk = H.sketches_from_arrays("reads.fq", 10**10, 8 * 10**9, kc, km, params, filt)
    small_kc, small_km = big_sketch(1000, 31, 6)
    sk.append(H.sketches_from_arrays("second", 5, 4, small_kc, small_km, params, H.FilterParams(False)))
    raw = sk.to_bsk()
    root = CM.Message(raw).root(CM.FINCH, "Multisketch")
    assert len(root["sketches"]) == 2
    check_bsk_against(root["sketches"][0], "reads.fq", (10**10, 8 * 10**9), kc, km, params, filt)
    check_bsk_against(root["sketches"][1], "second", (5, 4), small_kc, small_km, params, H.FilterParams(False))
    back = H.sketches_from_bsk(raw)
    assert len(back) == 2 and back.params_of(0) == params and back.params_of(1) == params
    L = H.lib()
    n = L.finch_sketch_n_hashes(back._p, 0)
    hs, cs, es, kk = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros((n, 31), np.uint8)
    assert L.finch_sketch_copy(back._p, 0, hs.ctypes.data, cs.ctypes.data, es.ctypes.data, kk.ctypes.data) == 0
    assert np.array_equal(hs, kc["hash"]) and np.array_equal(cs, kc["count"]) and np.array_equal(es, kc["extra_count"]) and np.array_equal(kk, km)
    fp = H.CFilterParams()
    assert L.finch_sketch_filter_params(back._p, 0, fp) == 0
    assert H.FilterParams.from_c(fp) == filt
    rawm = sk.to_msh()
    mm = CM.Message(rawm).root(CM.MASH, "MinHash")
    refs = mm["referenceList"]["references"]
    assert np.array_equal(refs[0]["hashes64"], kc["hash"]) and np.array_equal(refs[0]["counts32"], kc["count"])
    assert np.array_equal(refs[1]["hashes64"], small_kc["hash"]) and mm["minHashesPerWindow"] == 2_000_000 and mm["kmerSize"] == 31
    backm = H.sketches_from_msh(rawm)
    n = L.finch_sketch_n_hashes(backm._p, 0)
    hs, cs = np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    assert L.finch_sketch_copy(backm._p, 0, hs.ctypes.data, cs.ctypes.data, None, None) == 0
    assert np.array_equal(hs, kc["hash"]) and np.array_equal(cs, kc["count"])


def test_readers_take_multi_segment_messages_with_far_pointers(golden_dir):
    """the capnp crate's default allocator starts a new segment whenever the current one is full, so the reference's own
    .bsk / .msh files are multi-segment with far (and double-far) pointers; re-lay a message that way and read it back"""
    sk, kc, km, totals = golden_sketches(golden_dir, name="query")
    sk.set_comment(0, "a comment")
    raw = sk.to_bsk()
    w = np.frombuffer(raw, dtype="<u8", offset=8)
    S = 3
    want = H.sketches_from_bsk(raw).sketch(0)
    for ptr_word, double in ((S + 2, False), (S + 2, True), (S + 3, False), (S + 3, True)):  # name, comment
        moved = CM.far_split(raw, ptr_word, double)
        assert struct.unpack_from("<I", moved, 0)[0] == (2 if double else 1)
        got = H.sketches_from_bsk(moved)
        same_sketch(got.sketch(0), want)
        root = CM.Message(moved).root(CM.FINCH, "Multisketch")
        assert root["sketches"][0]["name"] == b"query" and root["sketches"][0]["comment"] == b"a comment"
    # a k-mer's Data moved behind a far pointer; then .msh with its hash list moved
    first_kmer_ptr = int(np.nonzero(w == kc["hash"][0])[0][0]) + 2
    got = H.sketches_from_bsk(CM.far_split(raw, first_kmer_ptr, True)).sketch(0)
    same_sketch(got, want)
    rawm = sk.to_msh()
    wm = np.frombuffer(rawm, dtype="<u8", offset=8)
    E = 1 + 7 + 1 + 1 + 1  # root pointer, MinHash, alphabet, ReferenceList, list tag -> first Reference (see the layout test)
    assert wm[E + 1] == totals[0]
    for double in (False, True):
        moved = CM.far_split(rawm, E + 3 + 5, double)  # hashes64 @6 = pointer 5
        got = H.sketches_from_msh(moved).sketch(0)
        assert np.array_equal(got.arrays[0]["hash"], kc["hash"]) and np.array_equal(got.arrays[0]["count"], kc["count"])
    # reference files that carry the sketches in referenceListOld (mash.rs:85-89): move the pointer over
    old = bytearray(rawm)
    base = 8 + 8 * 1
    new_ptr = struct.unpack_from("<Q", old, base + 8 * 6)[0]
    off = (new_ptr & 0xFFFFFFFF) >> 2
    struct.pack_into("<Q", old, base + 8 * 3, (new_ptr & ~0xFFFFFFFF) | ((off + 3) << 2))  # pointer 0 sits 3 words earlier
    struct.pack_into("<Q", old, base + 8 * 6, 0)
    got = H.sketches_from_msh(bytes(old)).sketch(0)
    assert np.array_equal(got.arrays[0]["hash"], kc["hash"])


def test_sk_reader_follows_serde_rules():
    base = {"kmer": 21, "alphabet": "ACGT", "preserveCase": False, "canonical": True, "sketchSize": 3,
            "hashType": "MurmurHash3_x64_128", "hashBits": 64, "hashSeed": 42, "scale": None,
            "sketches": [{"name": "aé\"\\\n", "seqLength": 10, "numValidKmers": 7, "comment": "c", "unknownKey": [1, {"x": None}],
                          "filters": {"minCopies": "2", "maxCopies": "90", "errFilter": "0.25", "strandFilter": "0.1"},
                          "hashes": ["1", "18446744073709551615", "7"], "kmers": ["AC", "GT", "TT"], "counts": [3, 4294967295, 1]}]}
    sk = H.sketches_from_json(json.dumps(base))
    s = sk.sketch(0)
    assert s.name == "aé\"\\\n" and (s.seq_length, s.num_valid_kmers, s.comment) == (10, 7, "c")
    assert s.arrays[0]["hash"].tolist() == [1, 2**64 - 1, 7] and s.arrays[0]["count"].tolist() == [3, 2**32 - 1, 1]
    assert s.arrays[0]["extra_count"].tolist() == [1, (2**32 - 1) // 2, 0]
    assert s.filter_params == H.FilterParams(True, (2, 90), 0.25, 0.1)
    p = sk.params_of(0)
    assert (p.kind, p.kmers_to_sketch, p.final_size, p.no_strict, p.kmer_length, p.hash_seed) == ("mash", 3, 3, True, 21, 42)
    # Option fields missing or null; kmers / counts absent -> empty k-mers, counts of 1 (json.rs:113-121)
    lean = dict(base, scale=0.001, sketches=[{"name": "x", "hashes": ["5"], "seqLength": None}])
    sk = H.sketches_from_json(json.dumps(lean).encode())
    s = sk.sketch(0)
    assert (s.seq_length, s.num_valid_kmers, s.comment) == (0, 0, "") and s.arrays[0]["count"].tolist() == [1]
    assert s.filter_params == H.FilterParams(False) and all(h.kmer == b"" for h in s.hashes)
    p = sk.params_of(0)
    assert (p.kind, p.kmers_to_sketch, p.scale) == ("scaled", 3, 0.001)
    del lean["scale"]  # Option<f64>: a missing key is None as well
    assert H.sketches_from_json(json.dumps(lean)).params_of(0).kind == "mash"
    assert H.sketches_from_json(json.dumps(dict(base, hashType="None", hashBits=0))).params_of(0).kind == "allcounts"
    # the reference's error texts (json.rs:163-199)
    with pytest.raises(FinchError, match=r"Multisketch has incompatible hash size \(32 != 64\)"):
        H.sketches_from_json(json.dumps(dict(base, hashBits=32)))
    with pytest.raises(FinchError, match="SHA1 sketch type is not supported"):
        H.sketches_from_json(json.dumps(dict(base, hashType="SHA1")))
    for broken in (dict(base, kmer=300), dict(base, kmer="21"), {k: v for k, v in base.items() if k != "alphabet"},
                   dict(base, sketches=[{"name": "x", "hashes": [5]}]), dict(base, sketches=[{"name": "x", "hashes": ["-1"]}]),
                   dict(base, sketches=[{"hashes": ["1"]}]), dict(base, sketches=[{"name": "x", "hashes": ["1", "2"], "counts": [1]}]),
                   dict(base, sketches=[{"name": "x", "hashes": ["1"], "counts": [-1]}]),
                   dict(base, sketches=[{"name": "x", "hashes": ["1"], "filters": {"minCopies": "two"}}])):
        with pytest.raises(FinchError):
            H.sketches_from_json(json.dumps(broken))
    for text in ("", "{", "[1,2]", json.dumps(base) + "x", '{"kmer": 01}', '{"a": "\\ud800"}'):
        with pytest.raises(FinchError):
            H.sketches_from_json(text)


def test_filter_sketch_updates_parameters_only(golden_dir):
    """FilterParams::filter_sketch (filtering.rs:20-54): stricter-of-both parameters; the hashes stay (the reference
    drops the filtered list it computes)"""
    sk, kc, km, _ = golden_sketches(golden_dir, filters=H.FilterParams(True, (2, 50), 0.1, 0.3))
    before = sk.sketch(0)
    sk.filter_sketch(0, H.FilterParams(True, (5, 80), 0.05, 0.4))
    after = sk.sketch(0)
    assert after.filter_params == H.FilterParams(True, (5, 50), 0.1, 0.4)
    assert np.array_equal(after.arrays[0], before.arrays[0])
    sk.filter_sketch(0, H.FilterParams(False, (None, 7), 0.0, 0.0))
    assert sk.sketch(0).filter_params == H.FilterParams(False, (None, 7), 0.1, 0.4)


def test_malformed_messages_are_errors_not_crashes(golden_dir):
    sk, kc, km, _ = golden_sketches(golden_dir)
    raw, rawm = sk.to_bsk(), sk.to_msh()
    for data in (b"", b"\0" * 7, raw[:8], raw[:40], raw[:-8], struct.pack("<II", 0, 10**6) + raw[8:], struct.pack("<II", 5000, 1)):
        with pytest.raises(FinchError):
            H.sketches_from_bsk(data)
        with pytest.raises(FinchError):
            H.sketches_from_msh(data)
    # lists that claim elements without bytes behind them: a void list / bit list / composite list of empty structs with
    # a count of 2^29 - 1 fits a 24 .. 40-byte message.  Errors -- the capnp runtime charges such elements against its
    # traversal limit -- never an allocation for half a billion sketches (which used to end in std::terminate)
    def seg(*words):
        body = b"".join(struct.pack("<Q", w) for w in words)
        return struct.pack("<II", 0, len(body) // 8) + body
    huge = (1 << 29) - 1
    root = 0 | (0 << 32) | (1 << 48)  # struct pointer: offset 0, no data words, one pointer
    bombs = [seg(root, 1 | (0 << 32) | (huge << 35)),                      # List(Void)
             seg(root, 1 | (1 << 32) | (7 << 35)),                         # List(Bool) -- cannot be a struct list
             seg(root, 1 | (7 << 32) | (0 << 35), (huge << 2))]            # composite, tag says 2^29-1 elements of 0 words
    for data in bombs:
        for fn in (H.sketches_from_bsk, H.sketches_from_msh):
            with pytest.raises(FinchError):
                got = fn(data)
                if len(got) == 0:  # (a message that decodes to nothing is fine too)
                    raise FinchError("empty")
    # primitive lists read as List(struct): the capnp runtime behind the reference's reader upgrades every encoding but the
    # bit list (fields beyond the element read as defaults) -- a List(UInt8) / List(UInt16) / small List(Void) where the
    # sketches belong is 16 / 4 / 5 default sketches, not an error
    for data, n in ((seg(root, 1 | (2 << 32) | (16 << 35), 0x0807060504030201, 0), 16),
                    (seg(root, 1 | (3 << 32) | (4 << 35), 0x0003000200010007), 4), (seg(root, 1 | (0 << 32) | (5 << 35)), 5)):
        got = H.sketches_from_bsk(data)
        assert len(got) == n
        for i in range(n):
            one = got.sketch(i)
            assert (one.name, one.seq_length, one.num_valid_kmers, len(one.arrays[0])) == ("", 0, 0, 0)
    # the wrong schema behind the right framing: whatever it decodes to, or a clean error -- never a crash
    for fn, blob in ((H.sketches_from_msh, raw), (H.sketches_from_bsk, rawm)):
        try:
            fn(blob)
        except FinchError:
            pass
    rng = np.random.default_rng(12)
    n_err = 0
    for blob in (raw, rawm):
        for _ in range(1500):
            b = bytearray(blob)
            for _ in range(int(rng.integers(1, 4))):
                i = int(rng.integers(0, len(b)))
                b[i] = int(rng.integers(0, 256)) if rng.random() < 0.5 else b[i] ^ (1 << int(rng.integers(0, 8)))
            for fn in (H.sketches_from_bsk, H.sketches_from_msh):
                try:
                    got = fn(bytes(b))
                    for i in range(len(got)):
                        got.sketch(i)
                except FinchError:
                    n_err += 1
    assert n_err > 100  # plenty of the mutations are caught as malformed; none may crash
    # incompatible sketches cannot share a Mash file (SketchParams::from_sketches, mod.rs:158-180)
    other = H.sketches_from_arrays("o", 1, 1, kc, km[:, :20].copy(), SketchParams.mash(10, 10, True, 20, 0), H.FilterParams(False))
    sk.append(other)
    with pytest.raises(FinchError, match="First sketch has k 21, but sketch 2 has k 20"):
        sk.to_msh()
    sk.to_bsk()  # write_finch_file stores parameters per sketch and does not compare them
