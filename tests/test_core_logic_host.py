"""Logic test of the per-lane kernel arithmetic (finch_rs_amd/csrc/fh_core.h) compiled for the HOST,
against the oracle.  This validates classification, rolling canonical selection and the LUT-based
murmur3 for every k in 1..32 without a GPU.  (The device build of the same header is checked by the
-m gpu parity tests.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostcore", "fhcore_host.cpp")
SO = os.path.join(HERE, "hostcore", "libfhcore_host.so")


@pytest.fixture(scope="module")
def core():
    hdr = os.path.join(HERE, "..", "finch_rs_amd", "csrc", "fh_core.h")
    if (not os.path.exists(SO)) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        # (several pytest-xdist workers may get here at once: each builds its own file and renames it into place)
        tmp = "%s.tmp.%d" % (SO, os.getpid())
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", tmp, SRC])
        os.replace(tmp, SO)
    L = C.CDLL(SO)
    L.fhcore_positions.restype = C.c_int
    L.fhcore_positions.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint64] + [C.c_void_p] * 4
    L.fhcore_positions_w.restype = C.c_int
    L.fhcore_positions_w.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint64] + [C.c_void_p] * 4
    return L


def naive(seq: bytes, k: int, seed: int):
    """per start position: (valid, is_rc, hash) following needletail canonical_kmers + hash_f"""
    n = len(seq)
    out = []
    norm = bytes(seq)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    for p in range(n):
        w = norm[p:p + k]
        if len(w) < k or any(b not in (65, 67, 71, 84) for b in w):
            out.append((0, 0, 0))
            continue
        rc = bytes(comp[b] for b in reversed(w))
        if w < rc:
            out.append((1, 0, O.hash_f(w, seed)))
        else:
            out.append((1, 1, O.hash_f(rc, seed)))
    return out


@pytest.mark.parametrize("k", list(range(1, 33)))
def test_positions_match_oracle(core, k):
    rng = np.random.default_rng(100 + k)
    alphabet = np.frombuffer(b"ACGTNacgtuU-.x\x00\xff", dtype=np.uint8)
    probs = np.array([20, 20, 20, 20, 1.5, 2, 2, 2, 2, 1, 1, .3, .3, .3, .3, .3])
    probs = probs / probs.sum()
    seq = rng.choice(alphabet, size=int(rng.integers(150, 400)), p=probs)
    seed = int(rng.integers(0, 2**63)) if k % 2 else 0
    n = len(seq)
    hashes = np.zeros(n, dtype=np.uint64)
    valid = np.zeros(n, dtype=np.uint8)
    isrc = np.zeros(n, dtype=np.uint8)
    canon = np.zeros(n, dtype=np.uint64)
    rc = core.fhcore_positions(seq.ctypes.data, n, k, seed, hashes.ctypes.data, valid.ctypes.data, isrc.ctypes.data,
                               canon.ctypes.data)
    assert rc == 0
    norm = O.normalize(bytes(seq))  # no whitespace in the alphabet above => same length
    assert len(norm) == n
    ref = naive(norm, k, seed)
    for p, (v, r, h) in enumerate(ref):
        assert valid[p] == v, (k, p)
        if v:
            assert isrc[p] == r, (k, p)
            assert int(hashes[p]) == h, (k, p)


@pytest.mark.parametrize("k", list(range(33, 65)))
def test_wide_positions_match_oracle(core, k):
    """K = 33..64 (two-word k-mers, WindowsW): valid / strand / hash per position and the canonical k-mer's 2K bits"""
    rng = np.random.default_rng(500 + k)
    alphabet = np.frombuffer(b"ACGTNacgtuU-.x\x00\xff", dtype=np.uint8)
    probs = np.array([30, 30, 30, 30, .4, 2, 2, 2, 2, 1, 1, .1, .1, .1, .1, .1])
    probs = probs / probs.sum()
    seq = rng.choice(alphabet, size=int(rng.integers(300, 700)), p=probs)
    if k % 2 == 0:  # plant a reverse-palindromic k-mer: the tie must report rc
        half = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=k // 2)
        comp = {65: 84, 67: 71, 71: 67, 84: 65}
        seq[100:100 + k] = np.concatenate([half, np.array([comp[int(b)] for b in half[::-1]], np.uint8)])
    seed = int(rng.integers(0, 2**63)) if k % 2 else 0
    n = len(seq)
    hashes, valid, isrc = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    canon = np.zeros(2 * n, np.uint64)
    assert core.fhcore_positions_w(seq.ctypes.data, n, k, seed, hashes.ctypes.data, valid.ctypes.data, isrc.ctypes.data,
                                   canon.ctypes.data) == 0
    norm = O.normalize(bytes(seq))
    assert len(norm) == n
    ref = naive(norm, k, seed)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    n_valid = 0
    for p, (v, r, h) in enumerate(ref):
        assert valid[p] == v, (k, p)
        if v:
            n_valid += 1
            assert isrc[p] == r and int(hashes[p]) == h, (k, p)
            w = norm[p:p + k]
            km = bytes(comp[b] for b in reversed(w)) if r else w
            val = 0
            for b in km:
                val = (val << 2) | b"ACGT".index(b)
            assert (int(canon[2 * p + 1]) << 64) | int(canon[2 * p]) == val, (k, p)
    assert n_valid > 20
    if k % 2 == 0:
        assert valid[100] == 1 and isrc[100] == 1


def test_quarter_octave_buckets(core):
    """the bucket index the sampling pre-pass histograms hashes by, and its inverse (bucket -> upper edge), are consistent"""
    core.fhcore_qoct_index.restype = C.c_uint32
    core.fhcore_qoct_index.argtypes = [C.c_uint64]
    core.fhcore_qoct_upper_edge.restype = C.c_uint64
    core.fhcore_qoct_upper_edge.argtypes = [C.c_uint32]
    rng = np.random.default_rng(8)
    prev = -1
    xs = sorted(set([1, 2, 3, 4, 5, 7, 8, 9, 2**20, 2**20 + 1, 2**63, 2**64 - 1] + [int(x) for x in rng.integers(1, 2**63, 2000)] +
                    [int(x) for x in rng.integers(1, 2**20, 500)]))
    for x in xs:
        q = core.fhcore_qoct_index(x)
        assert q >= prev
        prev = q
        assert x <= core.fhcore_qoct_upper_edge(q)
        assert q == 0 or x > core.fhcore_qoct_upper_edge(q - 1), (x, q)
    assert core.fhcore_qoct_index(2**64 - 1) == 255 and core.fhcore_qoct_upper_edge(255) == 2**64 - 1


def test_palindrome_reports_rc(core):
    seq = np.frombuffer(b"ACGTACGTAATT", dtype=np.uint8).copy()
    n = len(seq)
    out = [np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8),
           np.zeros(n, dtype=np.uint64)]
    core.fhcore_positions(seq.ctypes.data, n, 4, 0, *[a.ctypes.data for a in out])
    # ACGT and AATT are their own reverse complement: tie -> is_rc = true (canonical_kmers: `if fwd < rc`)
    assert out[1][0] == 1 and out[2][0] == 1
    assert out[1][8] == 1 and out[2][8] == 1


def test_high_word_prefilter_never_drops_a_candidate(core):
    """the hot loop rejects on hi((ka+kb)*M2)+1 <= hi(tau)+2 before it forms the 64-bit hash: that test must
    pass every position whose real hash is <= tau (carry out of the low words, wrap at 2^32-1, tau = max)"""
    core.fhcore_prefilter_check.restype = C.c_uint64
    core.fhcore_prefilter_check.argtypes = [C.c_void_p] * 3 + [C.c_uint64, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(7)
    n = 400000
    a = rng.integers(0, 2**64, n, dtype=np.uint64)
    b = rng.integers(0, 2**64, n, dtype=np.uint64)
    tau = rng.integers(0, 2**64, n, dtype=np.uint64)
    # adversarial quarter: sums that land just below / above tau, with and without a low-word carry
    q = n // 4
    tau[:q] = rng.integers(0, 2**40, q, dtype=np.uint64)  # small thresholds like a converged sketch
    M2 = np.uint64(0xc4ceb9fe1a85ec53)
    M2_INV = np.uint64(pow(0xc4ceb9fe1a85ec53, -1, 2**64))
    fin = lambda x: x ^ (x >> np.uint64(33))
    full = lambda ka, kb: (fin(ka * M2) + fin(kb * M2)).astype(np.uint64)  # uint64 arithmetic wraps
    target = rng.integers(0, 2**41, q, dtype=np.uint64)
    # choose kb so that the hash == target: fin is an involution (it only folds the high word into the low one)
    want = (target - fin(a[:q] * M2)).astype(np.uint64)
    b[:q] = fin(want) * M2_INV
    assert np.array_equal(full(a[:q], b[:q]), target)
    tau[q:q + 1000] = np.uint64(2**64 - 1)
    tau[q + 1000:q + 2000] = np.uint64(0xFFFFFFFF00000000)
    tau[q + 2000:q + 3000] = np.uint64(0xFFFFFFFEFFFFFFFF)
    n_pass = C.c_uint64(0)
    n_true = C.c_uint64(0)
    bad = core.fhcore_prefilter_check(a.ctypes.data, b.ctypes.data, tau.ctypes.data, n, C.byref(n_pass), C.byref(n_true))
    assert bad == 0
    h = full(a, b)
    assert n_true.value == int(np.count_nonzero(h <= tau)) > q // 4  # parts_hash agrees; the adversarial part hits
    # selectivity: besides the true hits the prefilter passes only hashes within two high-word steps of tau
    lim = (tau >> np.uint64(32)) + np.uint64(2)
    loose = np.count_nonzero(((h >> np.uint64(32)) <= lim) | ((h >> np.uint64(32)) >= np.uint64(0xFFFFFFFE)))
    assert n_true.value <= n_pass.value <= loose


def test_classification_every_byte_value_every_slot(core):
    """all 256 byte values in each of the 16 slots of a chunk, neighbours random: good bit iff the byte is one of
    ACGTacgtUu (needletail normalize + canonical_kmers: everything else breaks k-mers), code A0 C1 G2 T/U3"""
    core.fhcore_classify.restype = None
    core.fhcore_classify.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(3)
    chunks = rng.integers(0, 256, (256 * 16, 16), dtype=np.uint8)
    for v in range(256):
        for slot in range(16):
            chunks[v * 16 + slot, slot] = v
    flat = np.ascontiguousarray(chunks)
    codes = np.zeros(len(flat), dtype=np.uint32)
    good = np.zeros(len(flat), dtype=np.uint32)
    core.fhcore_classify(flat.ctypes.data, len(flat), codes.ctypes.data, good.ctypes.data)
    code_of = {ord(c): i for c, i in zip("ACGT", range(4))}
    code_of.update({ord(c): i for c, i in zip("acgt", range(4))})
    code_of[ord("U")] = code_of[ord("u")] = 3
    exp_good = np.zeros(len(flat), dtype=np.uint32)
    for s in range(16):
        col = flat[:, s]
        is_base = np.isin(col, list(code_of))
        exp_good |= is_base.astype(np.uint32) << np.uint32(s)
        want = np.array([code_of.get(int(b), 0) for b in col], dtype=np.uint32)
        got = (codes >> np.uint32(2 * s)) & np.uint32(3)
        assert np.array_equal(got[is_base], want[is_base]), s
    assert np.array_equal(good, exp_good)
    assert not np.any(good >> np.uint32(16))


def test_classification_every_pair_of_neighbouring_bytes(core):
    """the zero-byte test of classify4 borrows from a byte into the next one: all 65 536 pairs of neighbouring bytes, inside a
    dword, across dwords and across the two halves of the chunk the flags are gathered by"""
    core.fhcore_classify.restype = None
    core.fhcore_classify.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    bases = np.frombuffer(b"ACGTacgtUuNn\x00\xff -", dtype=np.uint8)
    a, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    is_base = np.zeros(256, dtype=bool)
    is_base[list(b"ACGTacgtUu")] = True
    for slot in (0, 1, 2, 3, 6, 7, 8, 11, 13, 14):
        chunks = bases[rng.integers(0, len(bases), (65536, 16))]
        chunks[:, slot] = a.ravel()
        chunks[:, slot + 1] = b.ravel()
        flat = np.ascontiguousarray(chunks)
        codes = np.zeros(len(flat), dtype=np.uint32)
        good = np.zeros(len(flat), dtype=np.uint32)
        core.fhcore_classify(flat.ctypes.data, len(flat), codes.ctypes.data, good.ctypes.data)
        exp = np.zeros(len(flat), dtype=np.uint32)
        for s_ in range(16):
            exp |= is_base[flat[:, s_]].astype(np.uint32) << np.uint32(s_)
        assert np.array_equal(good, exp), slot
