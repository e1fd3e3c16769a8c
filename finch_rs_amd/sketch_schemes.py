"""Host-side mirror of finch's sketch-scheme interface for the accelerated path.

Mirrors lib/src/sketch_schemes/mod.rs of the reference: `KmerCount` (16-22), `trait SketchScheme`
(24-51: process / total_bases_and_kmers / to_vec / parameters), `SketchParams` (54-128:
create_sketcher, process_post_filter, k, hash_info, expected_size).  The sketchers are thin handles
on the MI355X engine behind the C ABI (include/finch_hip.h); there is no CPU implementation here.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import FhParams, FinchHipError, KIND_MASH, KIND_SCALED, check

KC_DTYPE = np.dtype([("hash", "<u8"), ("count", "<u4"), ("extra_count", "<u4")])


class FinchError(Exception):
    """errors.rs: FinchError::Message"""


@dataclass
class KmerCount:
    """mod.rs:16-22"""
    hash: int
    kmer: bytes
    count: int
    extra_count: int
    label: Optional[bytes] = None


@dataclass
class SketchParams:
    """mod.rs:54-71.  kind: 'mash' | 'scaled' (AllCounts is not on the accelerated path)."""
    kind: str = "mash"
    kmers_to_sketch: int = 1000
    final_size: int = 1000
    no_strict: bool = False
    kmer_length: int = 21
    hash_seed: int = 0
    scale: float = 0.001

    @staticmethod
    def default():
        return SketchParams()  # mod.rs:73-83

    @staticmethod
    def mash(kmers_to_sketch=1000, final_size=1000, no_strict=False, kmer_length=21, hash_seed=0):
        return SketchParams("mash", kmers_to_sketch, final_size, no_strict, kmer_length, hash_seed)

    @staticmethod
    def scaled(kmers_to_sketch, kmer_length, scale, hash_seed=0):
        return SketchParams("scaled", kmers_to_sketch, kmers_to_sketch, False, kmer_length, hash_seed, scale)

    def k(self) -> int:
        return self.kmer_length

    def hash_info(self):  # mod.rs:138-146
        return ("MurmurHash3_x64_128", 64, self.hash_seed, self.scale if self.kind == "scaled" else None)

    def expected_size(self) -> int:  # mod.rs:148-156
        return self.final_size if self.kind == "mash" else self.kmers_to_sketch

    def create_sketcher(self, device: int = 0, **kw) -> "HipSketcher":
        """mod.rs:86-113 -> the device engine"""
        if self.kind == "mash":
            return HipSketcher(KIND_MASH, self.kmers_to_sketch, self.kmer_length, self.hash_seed, device=device, **kw)
        if self.kind == "scaled":
            return HipSketcher(KIND_SCALED, self.kmers_to_sketch, self.kmer_length, self.hash_seed, self.scale,
                               device=device, **kw)
        raise FinchError("sketch type %r is not on the accelerated path" % self.kind)

    def process_post_filter(self, kmers: List[KmerCount], name: str) -> List[KmerCount]:
        """mod.rs:115-128: Mash only -- truncate(final_size); error unless no_strict"""
        if self.kind == "mash":
            kmers = kmers[: self.final_size]
            if not self.no_strict and len(kmers) < self.final_size:
                raise FinchError("%s had too few kmers (%d) to sketch" % (name, len(kmers)))
        return kmers


class HipSketcher:
    """`impl SketchScheme` over the C ABI: MashSketcher (mash.rs) / ScaledSketcher (scaled.rs) on the GPU."""

    def __init__(self, kind: int, size: int, kmer_length: int, seed: int, scale: float = 0.001, device: int = 0,
                 max_launch: int = 0, hash_mask: int = 0, stage_bytes: int = 0):
        self._L = _lib.load()
        self.kind, self.size, self.kmer_length, self.seed, self.scale = kind, size, kmer_length, seed, scale
        self.device = device
        p = FhParams(kind, kmer_length, size, seed, scale, max_launch, hash_mask, stage_bytes)
        self._h = self._L.fh_new(C.byref(p), device)
        if not self._h:
            raise FinchHipError(-1, (self._L.fh_last_error() or b"").decode(errors="replace"))
        self.total_bases = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.fh_free(self._h)
            self._h = None

    __del__ = close

    # --- trait SketchScheme ---
    def process(self, seq) -> None:
        """process(&mut self, seq): one record's raw sequence() bytes (mash.rs:67-80)"""
        b = bytes(seq)
        self.total_bases += len(b)
        check(self._L.fh_process(self._h, C.cast(C.c_char_p(b), C.c_void_p), len(b)))

    def process_records(self, base: np.ndarray, offsets: np.ndarray, lens: np.ndarray) -> None:
        """fh_process for every record (base[offsets[i] : offsets[i] + lens[i]]) in one call: the loop a binding at the trait
        level runs, without Python in it"""
        base = np.ascontiguousarray(base, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
        if len(offsets) != len(lens):
            raise ValueError("records outside the buffer")
        # (the bounds of every record are checked where the loop is: two compares per record, not three passes of numpy)
        taken = C.c_uint64(0)
        rc = self._L.fh_process_records_in(self._h, base.ctypes.data, base.size, offsets.ctypes.data, lens.ctypes.data, len(lens),
                                           C.byref(taken))
        self.total_bases += taken.value
        if rc == _lib.FH_ERR_INVALID and b"outside the buffer" in (self._L.fh_last_error() or b""):
            raise ValueError("records outside the buffer")
        check(rc)

    def push_block(self, block) -> None:
        """several records at once: sequences separated/terminated by a breaker byte (0)"""
        a = np.ascontiguousarray(np.frombuffer(block, dtype=np.uint8) if not isinstance(block, np.ndarray) else block)
        check(self._L.fh_push_block(self._h, a.ctypes.data_as(C.c_void_p), a.size))

    def push_device(self, dev_ptr: int, nbytes: int) -> None:
        check(self._L.fh_push_device(self._h, C.c_void_p(dev_ptr), nbytes))

    def set_record_stride(self, stride: int) -> None:
        """records of the packed streams pushed from now on are `stride` - 1 bases long (0: not known, 1: not of one length);
        a tuning hint, never part of the result (include/finch_hip.h)"""
        check(self._L.fh_set_record_stride(self._h, stride))

    def debug_segments(self):
        """(launches of the segment kernel, blocks probed for a stride, stride of the last block)"""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        check(self._L.fh_debug_segments(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_stream_offset(self, off: int) -> None:
        check(self._L.fh_set_stream_offset(self._h, off))

    def reset(self) -> None:
        check(self._L.fh_reset(self._h))
        self.total_bases = 0

    def sync(self) -> None:
        check(self._L.fh_sync(self._h))

    def finish(self) -> Tuple[int, int]:
        n, tk = C.c_uint64(), C.c_uint64()
        check(self._L.fh_finish(self._h, C.byref(n), C.byref(tk)))
        return n.value, tk.value

    def total_bases_and_kmers(self) -> Tuple[int, int]:
        """mash.rs:82-84 (total_bases is a host counter; see include/finch_hip.h)"""
        return self.total_bases, self.finish()[1]

    def to_arrays(self, out=None):
        """-> (structured [hash,count,extra_count], kmers uint8 [n,k], first_pos uint64 [n]) ascending by hash.
        `out`: arrays of an earlier call to fill again (a caller that keeps its buffers: 2 M records are 110 MB, and first
        touching fresh pages costs several times what the copy does); used if they have room, views of them are returned."""
        n, _ = self.finish()
        def fits(a, dtype, ndim):  # the library writes through raw pointers: only arrays laid out as it expects
            return (isinstance(a, np.ndarray) and a.dtype == dtype and a.ndim == ndim and a.flags.c_contiguous and
                    a.flags.writeable and len(a) >= n)
        if (out is not None and len(out) == 3 and fits(out[0], KC_DTYPE, 1) and fits(out[1], np.uint8, 2) and
                out[1].shape[1] == self.kmer_length and fits(out[2], np.uint64, 1)):
            kc, km, ps = out[0][:n], out[1][:n], out[2][:n]
        else:
            kc = np.empty(n, dtype=KC_DTYPE)  # = struct fh_kmer_count
            km = np.empty((n, self.kmer_length), dtype=np.uint8)
            ps = np.empty(n, dtype=np.uint64)
        check(self._L.fh_copy_out_records(self._h, kc.ctypes.data_as(C.c_void_p), km.ctypes.data_as(C.c_void_p),
                                          ps.ctypes.data_as(C.c_void_p)))
        return kc, km, ps

    def to_vec(self) -> List[KmerCount]:
        """mash.rs:86-102"""
        kc, km, _ = self.to_arrays()
        return [KmerCount(int(kc["hash"][i]), bytes(km[i]), int(kc["count"][i]), int(kc["extra_count"][i]))
                for i in range(len(kc))]

    def parameters(self) -> SketchParams:
        """mash.rs:104-112 / scaled.rs:102-109"""
        if self.kind == KIND_MASH:
            return SketchParams("mash", self.size, self.size, False, self.kmer_length, self.seed)
        return SketchParams("scaled", self.size, self.size, False, self.kmer_length, self.seed, self.scale)

    def merge(self, other: "HipSketcher") -> None:
        check(self._L.fh_merge(self._h, other._h))

    def merge_arrays(self, kc, km, first_pos, total_kmers: int) -> None:
        hs = np.ascontiguousarray(kc["hash"], dtype=np.uint64)
        cs = np.ascontiguousarray(kc["count"], dtype=np.uint32)
        es = np.ascontiguousarray(kc["extra_count"], dtype=np.uint32)
        km = np.ascontiguousarray(km, dtype=np.uint8)
        ps = np.ascontiguousarray(first_pos, dtype=np.uint64)
        check(self._L.fh_merge_arrays(self._h, len(hs), hs.ctypes.data_as(C.c_void_p), cs.ctypes.data_as(C.c_void_p),
                                      es.ctypes.data_as(C.c_void_p), km.ctypes.data_as(C.c_void_p),
                                      ps.ctypes.data_as(C.c_void_p), total_kmers))

    def debug_add_counts(self, add_count: int, add_extra: int) -> None:
        """test hook: move the two counters of every hash held so far up (include/finch_hip.h, fh_debug_add_counts)"""
        check(self._L.fh_debug_add_counts(self._h, add_count, add_extra))

    def debug_counters(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self._L.fh_debug_counters(self._h, C.byref(a), C.byref(b), C.byref(c)))
        d, e = C.c_uint64(), C.c_uint64()
        check(self._L.fh_debug_speculation(self._h, C.byref(d), C.byref(e)))
        f, g, h = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self._L.fh_debug_fast_path(self._h, C.byref(f), C.byref(g), C.byref(h)))
        return {"launches": a.value, "relaunches": b.value, "big_prunes": c.value, "spec": d.value,
                "spec_second_pass": e.value, "spec_deferred": f.value, "spec_recovered": g.value, "fused_finishes": h.value}

    # --- measurement ---
    def set_profiling(self, on: bool) -> None:
        check(self._L.fh_set_profiling(self._h, 1 if on else 0))

    def kernel_time(self):
        ms, n, pos = C.c_double(), C.c_uint64(), C.c_uint64()
        check(self._L.fh_kernel_time(self._h, C.byref(ms), C.byref(n), C.byref(pos)))
        return ms.value, n.value, pos.value


class BatchSketcher:
    """Many sketches per launch (include/finch_hip.h, fh_batch_*): the packed streams of a batch of files sketched by one
    launch, finished by one epilogue launch, one synchronisation -- what a worker of sketch_files (lib.rs:29-49) does with
    the files it has staged.  Mash, 1..3000 hashes, k <= 32.  `sketch_many(blocks)` -> per block either
    (records, kmers, first_pos, total_kmers) or None ("not taken": sketch that block through a HipSketcher)."""

    def __init__(self, size: int, kmer_length: int, seed: int = 0, device: int = 0, max_files: int = 64,
                 stage_bytes: int = 64 << 20):
        self._L = _lib.load()
        self.size, self.kmer_length, self.seed, self.device = size, kmer_length, seed, device
        self.max_files = max_files
        p = FhParams(KIND_MASH, kmer_length, size, seed, 0.0, 0, 0, 0)
        self._h = self._L.fh_batch_new(C.byref(p), device, max_files, stage_bytes)
        if not self._h:
            raise FinchHipError(-1, (self._L.fh_last_error() or b"").decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None):
            self._L.fh_batch_free(self._h)
            self._h = None

    __del__ = close

    def stage(self, slot: int) -> np.ndarray:
        """the pinned staging buffer of `slot` as a writable uint8 array"""
        buf, cap = C.c_void_p(), C.c_uint64()
        check(self._L.fh_batch_stage(self._h, slot, C.byref(buf), C.byref(cap)))
        return np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(cap.value,))

    def submit(self, slot: int, offsets, lens, two_bit: bool = False) -> None:
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = np.ascontiguousarray(lens, dtype=np.uint64)
        fn = self._L.fh_batch_submit_packed if two_bit else self._L.fh_batch_submit
        check(fn(self._h, slot, o.ctypes.data, n.ctypes.data, len(o)))

    def packed_bytes(self, n_positions: int) -> int:
        """bytes a file of n_positions occupies in the two-bit form (fh_batch_submit_packed)"""
        return int(self._L.fh_batch_packed_bytes(n_positions))

    def pack(self, block: np.ndarray, region: np.ndarray) -> None:
        """write a packed byte stream's two-bit form into `region` (a slice of the staging buffer)"""
        block = np.ascontiguousarray(block, dtype=np.uint8)
        check(self._L.fh_batch_pack(block.ctypes.data, len(block), region.ctypes.data, len(region)))

    def wait(self, slot: int, n_files: int) -> np.ndarray:
        st = np.zeros(max(n_files, 1), dtype=np.uint8)
        check(self._L.fh_batch_wait(self._h, slot, st.ctypes.data))
        return st[:n_files]

    def result(self, slot: int, i: int):
        n, tk = C.c_uint64(), C.c_uint64()
        check(self._L.fh_batch_result(self._h, slot, i, C.byref(n), C.byref(tk)))
        kc = np.empty(n.value, dtype=KC_DTYPE)
        km = np.empty((n.value, self.kmer_length), dtype=np.uint8)
        check(self._L.fh_batch_copy_out_records(self._h, slot, i, kc.ctypes.data, km.ctypes.data))
        hs = np.empty(n.value, dtype=np.uint64)
        cs = np.empty(n.value, dtype=np.uint32)
        es = np.empty(n.value, dtype=np.uint32)
        km2 = np.empty((n.value, self.kmer_length), dtype=np.uint8)
        ps = np.empty(n.value, dtype=np.uint64)
        check(self._L.fh_batch_copy_out(self._h, slot, i, hs.ctypes.data, cs.ctypes.data, es.ctypes.data, km2.ctypes.data, ps.ctypes.data))
        assert np.array_equal(hs, kc["hash"]) and np.array_equal(cs, kc["count"]) and np.array_equal(es, kc["extra_count"]) and np.array_equal(km, km2)
        return kc, km, ps, tk.value

    def sketch_many(self, blocks, slot: int = 0, two_bit: bool = False):
        """blocks: packed streams (bytes / uint8 arrays) -> list of result tuples / None, in order; as many batches as it takes.
        two_bit: stage every block in the two-bit form (0.375 bytes per position on the link)"""
        out = []
        buf = self.stage(slot)
        i = 0
        blocks = [np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b for b in blocks]
        while i < len(blocks):
            offs, lens, pos = [], [], 0
            while i < len(blocks) and len(offs) < self.max_files:
                need = self.packed_bytes(len(blocks[i])) if two_bit else len(blocks[i])
                if pos + need > len(buf):
                    break
                if two_bit:
                    self.pack(blocks[i], buf[pos:pos + need])
                else:
                    buf[pos:pos + need] = blocks[i]
                offs.append(pos)
                lens.append(len(blocks[i]))
                pos = (pos + need + 63) & ~63
                i += 1
            if not offs:
                raise ValueError("block %d does not fit the staging buffer" % i)
            self.submit(slot, offs, lens, two_bit)
            st = self.wait(slot, len(offs))
            out += [self.result(slot, j) if st[j] == 0 else None for j in range(len(offs))]
        return out

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(self._L.fh_batch_counters(self._h, C.byref(a), C.byref(b)))
        return {"taken": a.value, "not_taken": b.value}

    def set_profiling(self, on: bool) -> None:
        check(self._L.fh_batch_set_profiling(self._h, 1 if on else 0))

    def kernel_time(self):
        ms, n, pos = C.c_double(), C.c_uint64(), C.c_uint64()
        check(self._L.fh_batch_kernel_time(self._h, C.byref(ms), C.byref(n), C.byref(pos)))
        return ms.value, n.value, pos.value


class DeviceBuffer:
    """device memory through the C ABI (no torch needed)"""

    def __init__(self, nbytes: int, device: int = 0):
        self._L = _lib.load()
        self.device, self.nbytes = device, nbytes
        p = C.c_void_p()
        check(self._L.fh_device_alloc(device, nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(arr)
        check(self._L.fh_copy_to_device(self.device, C.c_void_p(self.ptr + offset), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def download(self, nbytes: int, offset: int = 0) -> np.ndarray:
        out = np.zeros(nbytes, dtype=np.uint8)
        check(self._L.fh_copy_from_device(self.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + offset), nbytes))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            self._L.fh_device_free(self.device, C.c_void_p(self.ptr))
            self.ptr = None

    __del__ = free


def synth_genome_host(length: int, seed: int) -> np.ndarray:
    out = np.zeros(length, dtype=np.uint8)
    check(_lib.load().fh_synth_genome_host(out.ctypes.data_as(C.c_void_p), length, seed))
    return out


def synth_reads_host(genome: np.ndarray, first_read: int, n_reads: int, read_len: int, seed: int, sub_ppm: int,
                     n_ppm: int) -> np.ndarray:
    out = np.zeros(n_reads * (read_len + 1), dtype=np.uint8)
    g = np.ascontiguousarray(genome, dtype=np.uint8)
    check(_lib.load().fh_synth_reads_host(out.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), g.size,
                                          first_read, n_reads, read_len, seed, sub_ppm, n_ppm))
    return out


def _splitmix64(x: int) -> int:
    m = 2**64 - 1
    x = (x + 0x9E3779B97F4A7C15) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    return x ^ (x >> 31)


def synth_fasta_length(i: int, seed: int, scale: float = 1.0) -> int:
    """length of synthetic genome file i (SURVEY 8d M4, configs[4]): log-uniform 1..10 Mb from the per-file seed"""
    u = (_splitmix64(seed + 7919 * i) >> 11) / float(1 << 53)
    return max(1000, int(1e6 * 10.0 ** u * scale))


def synth_fasta_file(i: int, seed: int, scale: float = 1.0) -> bytes:
    """synthetic genome file i as FASTA text: one record, 70-column lines, header '>genome_<i> len=<L>'"""
    L = synth_fasta_length(i, seed, scale)
    g = synth_genome_host(L, seed + 1000003 * (i + 1))
    rows = (L + 69) // 70
    a = np.full((rows, 71), ord("\n"), np.uint8)
    gp = np.zeros(rows * 70, np.uint8)
    gp[:L] = g
    a[:, :70] = gp.reshape(rows, 70)
    last = L - (rows - 1) * 70
    return b">genome_%05d len=%d\n" % (i, L) + a.reshape(-1)[:(rows - 1) * 71 + last].tobytes() + b"\n"


def synth_genome_device(buf: DeviceBuffer, length: int, seed: int):
    check(_lib.load().fh_synth_genome_device(buf.device, C.c_void_p(buf.ptr), length, seed))


def synth_reads_device(out: DeviceBuffer, genome: DeviceBuffer, genome_len: int, first_read: int, n_reads: int,
                       read_len: int, seed: int, sub_ppm: int, n_ppm: int, out_offset: int = 0):
    check(_lib.load().fh_synth_reads_device(out.device, C.c_void_p(out.ptr + out_offset), C.c_void_p(genome.ptr),
                                            genome_len, first_read, n_reads, read_len, seed, sub_ppm, n_ppm))


def measure_read_bandwidth(buf: "DeviceBuffer", nbytes: int, reps: int = 3) -> float:
    """streaming-read GB/s of this box's HBM over a resident buffer (reporting only)"""
    out = C.c_double()
    check(_lib.load().fh_measure_read_bandwidth(buf.device, C.c_void_p(buf.ptr), nbytes, reps, C.byref(out)))
    return out.value


def device_count() -> int:
    return _lib.load().fh_device_count()
