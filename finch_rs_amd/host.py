"""Python binding of the C++ host layer (include/finch_host.h): finch::sketch_files / sketch_stream,
FilterParams, the `.sk` (Mash-JSON) writer -- mirrors lib/src/lib.rs:29-94 of the reference."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import FinchHipError
from .sketch_schemes import KC_DTYPE, FinchError, KmerCount, SketchParams


class CSketchParams(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("kmer_length", C.c_uint32), ("kmers_to_sketch", C.c_uint64),
                ("final_size", C.c_uint64), ("no_strict", C.c_uint32), ("pad", C.c_uint32),
                ("hash_seed", C.c_uint64), ("scale", C.c_double)]


class CFilterParams(C.Structure):
    _fields_ = [("filter_on", C.c_int32), ("has_abun_lo", C.c_uint32), ("abun_lo", C.c_uint32),
                ("has_abun_hi", C.c_uint32), ("abun_hi", C.c_uint32), ("pad", C.c_uint32),
                ("err_filter", C.c_double), ("strand_filter", C.c_double)]


@dataclass
class FilterParams:
    """filtering.rs:11-16"""
    filter_on: Optional[bool] = False
    abun_filter: Tuple[Optional[int], Optional[int]] = (None, None)
    err_filter: float = 0.0
    strand_filter: float = 0.0

    def to_c(self) -> CFilterParams:
        lo, hi = self.abun_filter
        return CFilterParams(-1 if self.filter_on is None else int(bool(self.filter_on)), lo is not None, lo or 0,
                             hi is not None, hi or 0, 0, self.err_filter, self.strand_filter)

    @staticmethod
    def from_c(c: CFilterParams) -> "FilterParams":
        return FilterParams(None if c.filter_on < 0 else bool(c.filter_on),
                            (c.abun_lo if c.has_abun_lo else None, c.abun_hi if c.has_abun_hi else None),
                            c.err_filter, c.strand_filter)


@dataclass
class Sketch:
    """serialization/mod.rs:46-55"""
    name: str
    seq_length: int
    num_valid_kmers: int
    comment: str
    hashes: List[KmerCount]
    filter_params: FilterParams
    sketch_params: SketchParams
    arrays: tuple = field(default=None, repr=False)  # (structured [hash,count,extra_count], kmers uint8[n,k])


_P = C.c_void_p
_SYMS = {
    "finch_last_error": (C.c_char_p, []),
    "finch_default_sketch_params": (None, [C.POINTER(CSketchParams)]),
    "finch_default_filter_params": (None, [C.POINTER(CFilterParams)]),
    "finch_sketch_files": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(CSketchParams), C.POINTER(CFilterParams),
                                     C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "finch_sketch_buffer": (C.c_int, [_P, C.c_uint64, C.c_char_p, C.POINTER(CSketchParams), C.POINTER(CFilterParams),
                                      C.c_int, C.POINTER(_P)]),
    "finch_sketch_file_sharded": (C.c_int, [C.c_char_p, C.POINTER(CSketchParams), C.POINTER(CFilterParams), C.POINTER(C.c_int),
                                            C.c_uint32, C.c_uint64, C.POINTER(_P)]),
    "finch_sketch_buffer_sharded": (C.c_int, [_P, C.c_uint64, C.c_char_p, C.POINTER(CSketchParams), C.POINTER(CFilterParams),
                                              C.POINTER(C.c_int), C.c_uint32, C.c_uint64, C.POINTER(_P)]),
    "finch_shard_probe": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, _P, _P, C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "finch_sketches_to_bsk": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "finch_sketches_to_msh": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "finch_free_bytes": (None, [_P]),
    "finch_sketches_from_bsk": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "finch_sketches_from_msh": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "finch_sketches_from_json": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "finch_open_sketch_file": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "finch_write_sketch_file": (C.c_int, [_P, C.c_char_p]),
    "finch_sketch_params_of": (C.c_int, [_P, C.c_uint32, C.POINTER(CSketchParams)]),
    "finch_sketch_comment": (C.c_char_p, [_P, C.c_uint32]),
    "finch_sketch_set_comment": (C.c_int, [_P, C.c_uint32, C.c_char_p]),
    "finch_sketches_append": (C.c_int, [_P, _P]),
    "finch_filter_sketch": (C.c_int, [_P, C.c_uint32, C.POINTER(CFilterParams)]),
    "finch_sketch_cardinality": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint64)]),
    "finch_sketch_hist": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "finch_sketch_from_sketcher": (C.c_int, [_P, C.c_char_p, C.c_uint64, C.c_int, C.POINTER(CSketchParams), C.POINTER(CFilterParams),
                                             C.POINTER(_P)]),
    "finch_sketches_free": (None, [_P]),
    "finch_sketches_len": (C.c_uint32, [_P]),
    "finch_sketch_name": (C.c_char_p, [_P, C.c_uint32]),
    "finch_sketch_seq_length": (C.c_uint64, [_P, C.c_uint32]),
    "finch_sketch_num_valid_kmers": (C.c_uint64, [_P, C.c_uint32]),
    "finch_sketch_n_hashes": (C.c_uint64, [_P, C.c_uint32]),
    "finch_sketch_filter_params": (C.c_int, [_P, C.c_uint32, C.POINTER(CFilterParams)]),
    "finch_sketch_copy": (C.c_int, [_P, C.c_uint32, _P, _P, _P, _P]),
    "finch_sketches_to_json": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "finch_free_string": (None, [_P]),
    "finch_sketches_from_arrays": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, _P, _P, _P, _P,
                                             C.POINTER(CSketchParams), C.POINTER(CFilterParams), C.POINTER(_P)]),
    "finch_apply_filters": (C.c_int, [_P, C.c_uint32, C.POINTER(CFilterParams)]),
    "finch_guess_filter_threshold": (C.c_uint32, [_P, C.c_uint64, C.c_double]),
    "finch_fastx_scan": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "finch_fasta_count_chunked": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "finch_read_file_probe": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "finch_source_probe": (C.c_int, [_P, C.c_uint64, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "finch_debug_device_inflate": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "finch_debug_device_gzip": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "finch_debug_kernel_times": (None, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "finch_debug_file_batch": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "finch_debug_fastq_host_strip": (C.c_uint64, []),
    "finch_fastq_strip_probe": (C.c_int, [_P, C.c_uint64, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64)]),
    "finch_fasta_two_bit_probe": (C.c_int, [_P, C.c_uint64, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64)]),
    "finch_gzip_probe": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint32)]),
    "finch_bgzf_batch_probe": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
}
class CDistance(C.Structure):
    _fields_ = [("containment", C.c_double), ("jaccard", C.c_double), ("mash_distance", C.c_double),
                ("common_hashes", C.c_uint64), ("total_hashes", C.c_uint64)]


_SYMS["finch_distance"] = (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, C.c_int, C.POINTER(CDistance)])
_SYMS["finch_raw_distance"] = (C.c_int, [_P, C.c_uint64, _P, C.c_uint64, C.c_double, C.POINTER(CDistance)])
_bound = None


def lib():
    global _bound
    if _bound is None:
        L = _lib.load()
        for name, (res, args) in _SYMS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _bound = L
    return _bound


def _check(rc):
    if rc != 0:
        msg = (lib().finch_last_error() or b"").decode(errors="replace")
        if rc == _lib.FH_ERR_INVALID:
            raise FinchError(msg)
        raise FinchHipError(rc, msg)


def _params_c(p: SketchParams) -> CSketchParams:
    kind = {"mash": 0, "scaled": 1}[p.kind]
    return CSketchParams(kind, p.kmer_length, p.kmers_to_sketch, p.final_size, int(p.no_strict), 0, p.hash_seed, p.scale)


class Sketches:
    """owning wrapper of a finch_sketches*"""

    def __init__(self, ptr, params: SketchParams):
        self._p, self.params = ptr, params

    def __del__(self):
        if getattr(self, "_p", None) and lib is not None:  # module globals are gone at interpreter shutdown
            lib().finch_sketches_free(self._p)
            self._p = None

    def __len__(self):
        return lib().finch_sketches_len(self._p)

    def params_of(self, i: int) -> SketchParams:
        c = CSketchParams()
        _check(lib().finch_sketch_params_of(self._p, i, C.byref(c)))
        kind = {0: "mash", 1: "scaled", 2: "allcounts"}[c.kind]
        # (only the Scaled variant has a scale; the dataclass default stands in elsewhere so that equal variants compare equal)
        return SketchParams(kind, c.kmers_to_sketch, c.final_size, bool(c.no_strict), c.kmer_length, c.hash_seed,
                            c.scale if kind == "scaled" else SketchParams().scale)

    def sketch(self, i: int) -> Sketch:
        L = lib()
        n = L.finch_sketch_n_hashes(self._p, i)
        if self.params is None:
            return self._sketch_loaded(i)
        k = self.params.kmer_length
        hs, cs, es = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        km = np.zeros((n, k), np.uint8)
        _check(L.finch_sketch_copy(self._p, i, hs.ctypes.data, cs.ctypes.data, es.ctypes.data, km.ctypes.data))
        fp = CFilterParams()
        _check(L.finch_sketch_filter_params(self._p, i, C.byref(fp)))
        kc = np.zeros(n, dtype=KC_DTYPE)
        kc["hash"], kc["count"], kc["extra_count"] = hs, cs, es
        hashes = [KmerCount(int(hs[j]), bytes(km[j]), int(cs[j]), int(es[j])) for j in range(n)]
        return Sketch(L.finch_sketch_name(self._p, i).decode(), L.finch_sketch_seq_length(self._p, i),
                      L.finch_sketch_num_valid_kmers(self._p, i), "", hashes, FilterParams.from_c(fp), self.params,
                      (kc, km))

    def _sketch_loaded(self, i: int) -> Sketch:
        """a sketch that came out of a file: k-mers may be absent (.msh) or of any length"""
        L = lib()
        n = L.finch_sketch_n_hashes(self._p, i)
        p = self.params_of(i)
        hs, cs, es = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        km = np.zeros((n, p.kmer_length), np.uint8)
        _check(L.finch_sketch_copy(self._p, i, hs.ctypes.data, cs.ctypes.data, es.ctypes.data, km.ctypes.data))
        fp = CFilterParams()
        _check(L.finch_sketch_filter_params(self._p, i, C.byref(fp)))
        kc = np.zeros(n, dtype=KC_DTYPE)
        kc["hash"], kc["count"], kc["extra_count"] = hs, cs, es
        has_kmers = bool(km.any())
        hashes = [KmerCount(int(hs[j]), bytes(km[j]) if has_kmers else b"", int(cs[j]), int(es[j])) for j in range(n)]
        return Sketch(L.finch_sketch_name(self._p, i).decode(), L.finch_sketch_seq_length(self._p, i),
                      L.finch_sketch_num_valid_kmers(self._p, i), L.finch_sketch_comment(self._p, i).decode(), hashes,
                      FilterParams.from_c(fp), p, (kc, km))

    def _bytes(self, fn) -> bytes:
        out, n = _P(), C.c_uint64()
        _check(fn(self._p, C.byref(out), C.byref(n)))
        try:
            return C.string_at(out.value, n.value)
        finally:
            lib().finch_free_bytes(out)

    def to_bsk(self) -> bytes:
        """write_finch_file (serialization/mod.rs:123-166)"""
        return self._bytes(lib().finch_sketches_to_bsk)

    def to_msh(self) -> bytes:
        """write_mash_file (serialization/mash.rs:12-58)"""
        return self._bytes(lib().finch_sketches_to_msh)

    def write(self, path: str) -> None:
        """.sk / .json, .bsk or .msh by file name (cli/src/main.rs:53-70)"""
        _check(lib().finch_write_sketch_file(self._p, path.encode()))

    def append(self, other: "Sketches") -> None:
        _check(lib().finch_sketches_append(self._p, other._p))

    def set_comment(self, i: int, comment: str) -> None:
        _check(lib().finch_sketch_set_comment(self._p, i, comment.encode()))

    def filter_sketch(self, i: int, filters: FilterParams) -> None:
        """FilterParams::filter_sketch (filtering.rs:20-54)"""
        c = filters.to_c()
        _check(lib().finch_filter_sketch(self._p, i, C.byref(c)))

    def cardinality(self, i: int) -> int:
        """statistics.rs:8-23"""
        out = C.c_uint64()
        _check(lib().finch_sketch_cardinality(self._p, i, C.byref(out)))
        return out.value

    def hist(self, i: int) -> np.ndarray:
        """statistics.rs:30-47"""
        n = C.c_uint64()
        _check(lib().finch_sketch_hist(self._p, i, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.uint64)
        _check(lib().finch_sketch_hist(self._p, i, out.ctypes.data, n.value, C.byref(n)))
        return out

    def to_list(self) -> List[Sketch]:
        return [self.sketch(i) for i in range(len(self))]

    def to_json(self) -> str:
        out, n = _P(), C.c_uint64()
        _check(lib().finch_sketches_to_json(self._p, C.byref(out), C.byref(n)))
        try:
            return C.string_at(out.value, n.value).decode()
        finally:
            lib().finch_free_string(out)

    def apply_filters(self, i: int, filters: FilterParams) -> FilterParams:
        c = filters.to_c()
        _check(lib().finch_apply_filters(self._p, i, C.byref(c)))
        return FilterParams.from_c(c)


def sketch_files(filenames: Sequence[str], sketch_params: SketchParams, filters: FilterParams,
                 devices: Optional[Sequence[int]] = None, n_threads: int = 0) -> Sketches:
    """finch::sketch_files (lib.rs:29-49)"""
    arr = (C.c_char_p * len(filenames))(*[f.encode() for f in filenames])
    devs = list(devices) if devices else [0]
    darr = (C.c_int * len(devs))(*devs)
    sp, fp = _params_c(sketch_params), filters.to_c()
    out = _P()
    _check(lib().finch_sketch_files(arr, len(filenames), C.byref(sp), C.byref(fp), darr, len(devs), n_threads, C.byref(out)))
    return Sketches(out, sketch_params)


def sketch_stream(data: bytes, name: str, sketch_params: SketchParams, filters: FilterParams, device: int = 0) -> Sketches:
    """finch::sketch_stream (lib.rs:51-94) over an in-memory file image"""
    sp, fp = _params_c(sketch_params), filters.to_c()
    out = _P()
    buf = np.frombuffer(data, dtype=np.uint8)
    _check(lib().finch_sketch_buffer(buf.ctypes.data, len(data), name.encode(), C.byref(sp), C.byref(fp), device, C.byref(out)))
    return Sketches(out, sketch_params)


def sketch_file_sharded(filename: str, sketch_params: SketchParams, filters: FilterParams, devices: Sequence[int],
                        chunk_bytes: int = 0) -> Sketches:
    """ONE input partitioned over the handles in `devices` (an entry per handle; entries may repeat), partial sketches
    merged on the host: the north_star's single-large-input path (include/finch_host.h)"""
    devs = list(devices)
    darr = (C.c_int * len(devs))(*devs)
    sp, fp = _params_c(sketch_params), filters.to_c()
    out = _P()
    _check(lib().finch_sketch_file_sharded(filename.encode(), C.byref(sp), C.byref(fp), darr, len(devs), chunk_bytes, C.byref(out)))
    return Sketches(out, sketch_params)


def sketch_stream_sharded(data: bytes, name: str, sketch_params: SketchParams, filters: FilterParams, devices: Sequence[int],
                          chunk_bytes: int = 0) -> Sketches:
    devs = list(devices)
    darr = (C.c_int * len(devs))(*devs)
    sp, fp = _params_c(sketch_params), filters.to_c()
    out = _P()
    buf = np.frombuffer(data, dtype=np.uint8)
    _check(lib().finch_sketch_buffer_sharded(buf.ctypes.data, len(data), name.encode(), C.byref(sp), C.byref(fp), darr, len(devs),
                                             chunk_bytes, C.byref(out)))
    return Sketches(out, sketch_params)


def shard_probe(data: bytes, k: int, chunk_bytes: int):
    """chunks the sharded reader deals out (test hook, no device): [(text_off, length, start_state, halo bytes)], records, total_bases"""
    src = np.frombuffer(data, dtype=np.uint8)
    cap = len(data) // max(1, chunk_bytes // 4) + 64
    meta, halos = np.zeros(4 * cap, np.uint64), np.zeros(64 * cap, np.uint8)
    n, nr, tb = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _check(lib().finch_shard_probe(src.ctypes.data, len(data), k, chunk_bytes, cap, meta.ctypes.data, halos.ctypes.data,
                                   C.byref(n), C.byref(nr), C.byref(tb)))
    assert n.value <= cap
    out = []
    for i in range(n.value):
        off, ln, st, hl = (int(x) for x in meta[4 * i:4 * i + 4])
        out.append((off, ln, st, halos[64 * i:64 * i + hl].tobytes()))
    return out, nr.value, tb.value


def _loaded(fn, *args) -> Sketches:
    out = _P()
    _check(fn(*args, C.byref(out)))
    return Sketches(out, None)  # parameters live per sketch (Sketches.params_of)


def sketches_from_bsk(data: bytes) -> Sketches:
    """read_finch_file (serialization/mod.rs:168-222)"""
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    return _loaded(lib().finch_sketches_from_bsk, buf.ctypes.data, len(data))


def sketches_from_msh(data: bytes) -> Sketches:
    """read_mash_file (serialization/mash.rs:60-135)"""
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    return _loaded(lib().finch_sketches_from_msh, buf.ctypes.data, len(data))


def sketches_from_json(text) -> Sketches:
    """MultiSketch::to_sketches (serialization/json.rs:244-262)"""
    data = text.encode() if isinstance(text, str) else bytes(text)
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    return _loaded(lib().finch_sketches_from_json, buf.ctypes.data, len(data))


def open_sketch_file(path: str) -> Sketches:
    """finch::open_sketch_file (lib.rs:96-118)"""
    return _loaded(lib().finch_open_sketch_file, path.encode())


def sketch_from_sketcher(sketcher, name: str, seq_length: int, fmt: int, sketch_params: SketchParams, filters: FilterParams) -> Sketches:
    """the tail of sketch_stream (lib.rs:70-93) on a HipSketcher the caller fed itself: to_vec -> filters -> post filter"""
    sp, fp = _params_c(sketch_params), filters.to_c()
    out = _P()
    _check(lib().finch_sketch_from_sketcher(sketcher._h, name.encode(), seq_length, fmt, C.byref(sp), C.byref(fp), C.byref(out)))
    return Sketches(out, sketch_params)


def sketches_from_arrays(name, seq_length, num_valid_kmers, kc, km, sketch_params: SketchParams, filters: FilterParams) -> Sketches:
    hs = np.ascontiguousarray(kc["hash"], np.uint64)
    cs = np.ascontiguousarray(kc["count"], np.uint32)
    es = np.ascontiguousarray(kc["extra_count"], np.uint32)
    km = np.ascontiguousarray(km, np.uint8)
    sp, fp = _params_c(sketch_params), filters.to_c()
    out = _P()
    _check(lib().finch_sketches_from_arrays(name.encode(), seq_length, num_valid_kmers, len(hs), hs.ctypes.data, cs.ctypes.data,
                                            es.ctypes.data, km.ctypes.data if km.size else None, C.byref(sp), C.byref(fp),
                                            C.byref(out)))
    return Sketches(out, sketch_params)


def guess_filter_threshold(counts, level: float) -> int:
    a = np.ascontiguousarray(counts, np.uint32)
    return lib().finch_guess_filter_threshold(a.ctypes.data, len(a), level)


def fastx_scan(data: bytes):
    n, tb, fmt = C.c_uint64(), C.c_uint64(), C.c_int()
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    _check(lib().finch_fastx_scan(buf.ctypes.data, len(data), C.byref(n), C.byref(tb), C.byref(fmt)))
    return n.value, tb.value, fmt.value


def fasta_count_chunked(data: bytes, chunk: int):
    """(records, total_bases) as the device-side FASTA path counts them, text fed in `chunk`-byte pieces"""
    n, tb = C.c_uint64(), C.c_uint64()
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    _check(lib().finch_fasta_count_chunked(buf.ctypes.data, len(data), chunk, C.byref(n), C.byref(tb)))
    return n.value, tb.value


def bgzf_batch_probe(data: bytes, buf_bytes: int, max_members: int, text_budget: int, cap: int):
    """(text, batches, first byte) as the device-inflate reader would deal a BGZF image out, inflated on the host (test hook)"""
    src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    out = np.zeros(max(cap, 1), np.uint8)
    n, nb, fb = C.c_uint64(), C.c_uint64(), C.c_int()
    _check(lib().finch_bgzf_batch_probe(src.ctypes.data, len(data), buf_bytes, max_members, text_budget, out.ctypes.data, cap,
                                        C.byref(n), C.byref(nb), C.byref(fb)))
    return out[:n.value].tobytes(), nb.value, fb.value


def debug_device_inflate():
    """(inputs sketched with the BGZF inflate on the device, inputs re-read through the host inflate) -- test hook"""
    a, b = C.c_uint64(), C.c_uint64()
    lib().finch_debug_device_inflate(C.byref(a), C.byref(b))
    return a.value, b.value


def gzip_probe(data: bytes, piece_bytes: int = 1 << 20):
    """(header length, first text byte or -1, DEFLATE bytes the reader hands over, their CRC-32) of a gzip image -- test hook, no device"""
    src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    h, fb, n, crc = C.c_uint64(), C.c_int(), C.c_uint64(), C.c_uint32()
    _check(lib().finch_gzip_probe(src.ctypes.data, len(data), piece_bytes, C.byref(h), C.byref(fb), C.byref(n), C.byref(crc)))
    return h.value, fb.value, n.value, crc.value


def debug_device_gzip():
    """(inputs sketched with plain gzip inflated on the device, inputs re-read through the host inflate) -- test hook"""
    a, b = C.c_uint64(), C.c_uint64()
    lib().finch_debug_device_gzip(C.byref(a), C.byref(b))
    return a.value, b.value


def fastq_strip_probe(text: bytes, threads: int):
    """-> (packed stream, records, total_bases) of plain 4-line FASTQ text through the host-side strip; FinchError if it is not"""
    src = np.frombuffer(text, dtype=np.uint8)
    out = np.zeros(len(text) // 2 + 64, dtype=np.uint8)
    m, nr, nb = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _check(lib().finch_fastq_strip_probe(src.ctypes.data, len(text), threads, out.ctypes.data, len(out), C.byref(m), C.byref(nr), C.byref(nb)))
    return out[:m.value].tobytes(), nr.value, nb.value


def fasta_two_bit_probe(text: bytes, piece: int):
    """-> (region, positions, records, total_bases): FASTA text through the workers' piecewise walk into the two-bit form"""
    src = np.frombuffer(text, dtype=np.uint8)
    cap = ((len(text) + 2047) // 2048 + 1) * 768
    region = np.full(cap + 64, 0xA5, dtype=np.uint8)
    m, nr, nb = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _check(lib().finch_fasta_two_bit_probe(src.ctypes.data, len(text), piece, region.ctypes.data, cap, C.byref(m), C.byref(nr), C.byref(nb)))
    assert np.all(region[cap:] == 0xA5)
    return region[:((m.value + 2047) // 2048 + 1) * 768], m.value, nr.value, nb.value


def debug_fastq_host_strip() -> int:
    return lib().finch_debug_fastq_host_strip()


def debug_file_batch():
    """(files sketched many-per-launch, files the batch path handed to a sketcher of their own) by this process so far"""
    a, b = C.c_uint64(), C.c_uint64()
    lib().finch_debug_file_batch(C.byref(a), C.byref(b))
    return a.value, b.value


def debug_kernel_times(enable: int = -1):
    """(sketch-kernel ms, launches, positions) over the inputs sketched since the hook was switched on; then enable = 1 switches
    it on and zeroes the sums, 0 off, -1 leaves it -- measurement hook"""
    ms, nl, npos = C.c_double(), C.c_uint64(), C.c_uint64()
    lib().finch_debug_kernel_times(enable, C.byref(ms), C.byref(nl), C.byref(npos))
    return ms.value, nl.value, npos.value


def source_probe(data: bytes, chunk: int, cap: int) -> bytes:
    """the (decompressed) byte stream the parsers see for an input image, read `chunk` bytes at a time (test hook)"""
    src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    buf = np.zeros(max(cap, 1), np.uint8)
    got = C.c_uint64()
    _check(lib().finch_source_probe(src.ctypes.data, len(data), chunk, buf.ctypes.data, cap, C.byref(got)))
    return buf[:got.value].tobytes()


def read_file_probe(path: str, chunk: int, read_threads: int, cap: int) -> bytes:
    """the bytes the text paths get from a plain file read in `chunk`-byte requests (test hook)"""
    buf = np.zeros(max(cap, 1), np.uint8)
    got = C.c_uint64()
    _check(lib().finch_read_file_probe(path.encode(), chunk, read_threads, buf.ctypes.data, cap, C.byref(got)))
    return buf[:got.value].tobytes()


def raw_distance(query_hashes, ref_hashes, scale: float = 0.0):
    """distance.rs:66-126 -> (containment, jaccard, common, total)"""
    q = np.ascontiguousarray(query_hashes, np.uint64)
    r = np.ascontiguousarray(ref_hashes, np.uint64)
    d = CDistance()
    _check(lib().finch_raw_distance(q.ctypes.data if len(q) else None, len(q), r.ctypes.data if len(r) else None, len(r), scale, C.byref(d)))
    return d.containment, d.jaccard, d.common_hashes, d.total_hashes


def distance(a: Sketches, ia: int, b: Sketches, ib: int, old_mode: bool = False):
    """distance.rs:9-47 -> dict(containment, jaccard, mash_distance, common_hashes, total_hashes)"""
    d = CDistance()
    _check(lib().finch_distance(a._p, ia, b._p, ib, int(old_mode), C.byref(d)))
    return {"containment": d.containment, "jaccard": d.jaccard, "mash_distance": d.mash_distance,
            "common_hashes": d.common_hashes, "total_hashes": d.total_hashes}
