"""ctypes binding of the CPU oracle (oracle/libfinch_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py -- never by the product package (finch_rs_amd).  See oracle/finch_oracle.h for the
reference file:line each entry point restates and for the parity-pinning statement.
"""
import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfinch_oracle.so")

MASH, SCALED = 0, 1


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Idempotent."""
    src = os.path.join(_HERE, "finch_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _SO


def use_native() -> str:
    """Compile the oracle on THIS host with -march=native and bind that build (bench.py's cpu_baseline leg: the CPU
    baseline is timed with the flags BASELINE.md names).  Falls back to the portable build if the compiler is missing.
    Returns the flags in use."""
    global _lib, _SO
    native = os.path.join(_HERE, "libfinch_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])
    except Exception:
        lib()
        return "-O3 -march=x86-64-v2"
    _SO, _lib = native, None
    lib()
    return "-O3 -march=native"


class KmerCountC(C.Structure):
    _fields_ = [("hash", C.c_uint64), ("count", C.c_uint32), ("extra_count", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _SO.endswith("_native.so"):
            build()
        L = C.CDLL(_SO)
        L.fo_hash_f.restype = C.c_uint64
        L.fo_hash_f.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        L.fo_murmur3_x64_128.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64)]
        L.fo_normalize.restype = C.c_size_t
        L.fo_normalize.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.fo_reverse_complement.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.fo_new.restype = C.c_void_p
        L.fo_new.argtypes = [C.c_int, C.c_size_t, C.c_double, C.c_uint8, C.c_uint64]
        L.fo_free.argtypes = [C.c_void_p]
        L.fo_set_hash_mask.argtypes = [C.c_void_p, C.c_uint64]
        L.fo_push.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint8]
        L.fo_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.fo_process_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint8]
        L.fo_totals.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.fo_max_hash.restype = C.c_uint64
        L.fo_max_hash.argtypes = [C.c_void_p]
        L.fo_len.restype = C.c_size_t
        L.fo_len.argtypes = [C.c_void_p]
        L.fo_to_vec.restype = C.c_size_t
        L.fo_to_vec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.fo_sketch_stream.restype = C.c_int
        L.fo_sketch_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.fo_filter_strands.restype = C.c_size_t
        L.fo_filter_strands.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_double,
                                        C.c_void_p, C.c_void_p]
        L.fo_guess_filter_threshold.restype = C.c_uint32
        L.fo_guess_filter_threshold.argtypes = [C.c_void_p, C.c_size_t, C.c_double]
        L.fo_filter_abundance.restype = C.c_size_t
        L.fo_filter_abundance.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint32,
                                          C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


KC_DTYPE = np.dtype([("hash", "<u8"), ("count", "<u4"), ("extra_count", "<u4")])


def hash_f(item: bytes, seed: int = 0) -> int:
    return lib().fo_hash_f(item, len(item), seed)


def murmur3_x64_128(item: bytes, seed: int = 0) -> Tuple[int, int]:
    out = (C.c_uint64 * 2)()
    lib().fo_murmur3_x64_128(item, len(item), seed, out)
    return out[0], out[1]


def normalize(seq: bytes) -> bytes:
    out = C.create_string_buffer(max(1, len(seq)))
    n = lib().fo_normalize(seq, len(seq), out)
    return out.raw[:n]


def reverse_complement(seq: bytes) -> bytes:
    out = C.create_string_buffer(max(1, len(seq)))
    lib().fo_reverse_complement(seq, len(seq), out)
    return out.raw[: len(seq)]


def _as_buf(data):
    """bytes / bytearray / numpy uint8 -> (pointer, length, keepalive)"""
    if isinstance(data, np.ndarray):
        a = np.ascontiguousarray(data, dtype=np.uint8)
        return a.ctypes.data_as(C.c_void_p), a.size, a
    b = bytes(data)
    return C.cast(C.c_char_p(b), C.c_void_p), len(b), b


class OracleSketcher:
    """MashSketcher (mash.rs) / ScaledSketcher (scaled.rs) restated on the CPU."""

    def __init__(self, kind: int = MASH, size: int = 1000, k: int = 21, seed: int = 0, scale: float = 0.001):
        self.kind, self.size, self.k, self.seed, self.scale = kind, size, k, seed, scale
        self._h = lib().fo_new(kind, size, float(scale), k, seed)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fo_free(self._h)
            self._h = None

    def set_hash_mask(self, mask: int):
        lib().fo_set_hash_mask(self._h, mask)

    def push(self, kmer: bytes, extra_count: int = 0):
        lib().fo_push(self._h, kmer, len(kmer), extra_count)

    def process(self, seq):
        p, n, keep = _as_buf(seq)
        lib().fo_process(self._h, p, n)

    def process_packed(self, buf, sep: int = 0):
        p, n, keep = _as_buf(buf)
        lib().fo_process_packed(self._h, p, n, sep)

    def sketch_stream(self, filebytes) -> int:
        p, n, keep = _as_buf(filebytes)
        return lib().fo_sketch_stream(self._h, p, n)

    def total_bases_and_kmers(self) -> Tuple[int, int]:
        a, b = C.c_uint64(), C.c_uint64()
        lib().fo_totals(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    @property
    def max_hash(self) -> int:
        return lib().fo_max_hash(self._h)

    def to_vec(self):
        """-> (structured array [hash,count,extra_count] ascending by hash, kmers uint8 [n,k])"""
        n = lib().fo_len(self._h)
        kc = np.zeros(n, dtype=KC_DTYPE)
        km = np.zeros((n, self.k), dtype=np.uint8)
        lib().fo_to_vec(self._h, kc.ctypes.data_as(C.c_void_p), km.ctypes.data_as(C.c_void_p))
        return kc, km


def filter_strands(kc: np.ndarray, kmers: Optional[np.ndarray], ratio: float):
    n = len(kc)
    k = kmers.shape[1] if kmers is not None else 0
    out = np.zeros(n, dtype=KC_DTYPE)
    kout = np.zeros((n, k), dtype=np.uint8)
    kc = np.ascontiguousarray(kc)
    m = lib().fo_filter_strands(kc.ctypes.data_as(C.c_void_p),
                                kmers.ctypes.data_as(C.c_void_p) if kmers is not None else None, n, k, ratio,
                                out.ctypes.data_as(C.c_void_p), kout.ctypes.data_as(C.c_void_p))
    return out[:m], kout[:m]


def guess_filter_threshold(kc: np.ndarray, level: float) -> int:
    kc = np.ascontiguousarray(kc)
    return lib().fo_guess_filter_threshold(kc.ctypes.data_as(C.c_void_p), len(kc), level)


def filter_abundance(kc: np.ndarray, kmers: Optional[np.ndarray], lo: Optional[int], hi: Optional[int]):
    n = len(kc)
    k = kmers.shape[1] if kmers is not None else 0
    out = np.zeros(n, dtype=KC_DTYPE)
    kout = np.zeros((n, k), dtype=np.uint8)
    kc = np.ascontiguousarray(kc)
    m = lib().fo_filter_abundance(kc.ctypes.data_as(C.c_void_p),
                                  kmers.ctypes.data_as(C.c_void_p) if kmers is not None else None, n, k,
                                  lo is not None, lo or 0, hi is not None, hi or 0,
                                  out.ctypes.data_as(C.c_void_p), kout.ctypes.data_as(C.c_void_p))
    return out[:m], kout[:m]
