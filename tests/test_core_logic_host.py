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
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    L = C.CDLL(SO)
    L.fhcore_positions.restype = C.c_int
    L.fhcore_positions.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint64] + [C.c_void_p] * 4
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


def test_palindrome_reports_rc(core):
    seq = np.frombuffer(b"ACGTACGTAATT", dtype=np.uint8).copy()
    n = len(seq)
    out = [np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8),
           np.zeros(n, dtype=np.uint64)]
    core.fhcore_positions(seq.ctypes.data, n, 4, 0, *[a.ctypes.data for a in out])
    # ACGT and AATT are their own reverse complement: tie -> is_rc = true (canonical_kmers: `if fwd < rc`)
    assert out[1][0] == 1 and out[2][0] == 1
    assert out[1][8] == 1 and out[2][8] == 1
