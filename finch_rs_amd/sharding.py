"""Read-block sharding of one large input across the GPUs of a node, and the host-side merge of the
tiny partial sketches (north_star: "no RCCL collective is needed").

The reference has no multi-device path; the property that makes sharding exact is that a Mash/Scaled
sketch is a function of the multiset of k-mers (mash.rs:34-63): the global bottom-n is the bottom-n of
the union of the shard sketches with counts summed.  Each rank sketches whole reads [lo, hi) of the
input; rank 0 gathers the <= n-record partial sketches as one fixed-size tensor per rank (a control-plane
gather of a few tens of KB, not a data-path collective) and merges them on the host.
"""
import ctypes as C
from typing import List, Tuple

import numpy as np

from . import _lib
from ._lib import check
from .sketch_schemes import KC_DTYPE, SketchParams


def shard_bounds(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous block of whole reads for `rank` (strong split of n_units over `world`)"""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def payload_words(pad_n: int, k: int) -> int:
    return 2 + pad_n * (4 + (k + 7) // 8)


def pack_partial(kc: np.ndarray, km: np.ndarray, pos: np.ndarray, total_kmers: int, pad_n: int, k: int) -> np.ndarray:
    """partial sketch -> fixed-size int64 vector (so that a plain gather works)"""
    n = len(kc)
    if n > pad_n:
        raise ValueError("partial sketch (%d) larger than the padded size (%d)" % (n, pad_n))
    kmw = (k + 7) // 8
    p = np.zeros(payload_words(pad_n, k), dtype=np.int64)
    p[0], p[1] = n, total_kmers
    off = 2
    p[off:off + n] = np.ascontiguousarray(kc["hash"]).view(np.int64); off += pad_n
    p[off:off + n] = kc["count"].astype(np.int64); off += pad_n
    p[off:off + n] = kc["extra_count"].astype(np.int64); off += pad_n
    p[off:off + n] = np.ascontiguousarray(pos, dtype=np.uint64).view(np.int64); off += pad_n
    kmp = np.zeros((pad_n, kmw * 8), dtype=np.uint8)
    kmp[:n, :k] = km
    p[off:off + pad_n * kmw] = kmp.view(np.int64).reshape(-1)
    return p


def unpack_partial(p: np.ndarray, pad_n: int, k: int):
    p = np.ascontiguousarray(p, dtype=np.int64)
    n, tk = int(p[0]), int(p[1])
    kmw = (k + 7) // 8
    off = 2
    kc = np.zeros(n, dtype=KC_DTYPE)
    kc["hash"] = p[off:off + n].view(np.uint64); off += pad_n
    kc["count"] = p[off:off + n].astype(np.uint32); off += pad_n
    kc["extra_count"] = p[off:off + n].astype(np.uint32); off += pad_n
    pos = p[off:off + n].view(np.uint64).copy(); off += pad_n
    km = p[off:off + pad_n * kmw].view(np.uint8).reshape(pad_n, kmw * 8)[:n, :k].copy()
    return kc, km, pos, tk


def merge_partials(params: SketchParams, partials: List[tuple]):
    """host-side merge (fh_merge_partials): [(kc, km, pos, total_kmers), ...] -> (kc, km, pos, total_kmers)"""
    L = _lib.load()
    kind = {"mash": 0, "scaled": 1}[params.kind]
    k = params.kmer_length
    kc, km, pos, tk = partials[0]
    for (kc2, km2, pos2, tk2) in partials[1:]:
        nA, nB = len(kc), len(kc2)
        cap = nA + nB
        oh, oc, oe = np.zeros(cap, np.uint64), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        ok, op = np.zeros((cap, k), np.uint8), np.zeros(cap, np.uint64)
        n_out = C.c_uint64()

        def a(x, dt):
            return np.ascontiguousarray(x, dtype=dt)
        hA, cA, eA, kA, pA = a(kc["hash"], np.uint64), a(kc["count"], np.uint32), a(kc["extra_count"], np.uint32), a(km, np.uint8), a(pos, np.uint64)
        hB, cB, eB, kB, pB = a(kc2["hash"], np.uint64), a(kc2["count"], np.uint32), a(kc2["extra_count"], np.uint32), a(km2, np.uint8), a(pos2, np.uint64)
        check(L.fh_merge_partials(kind, params.kmers_to_sketch, params.scale, k,
                                  nA, hA.ctypes.data, cA.ctypes.data, eA.ctypes.data, kA.ctypes.data, pA.ctypes.data,
                                  nB, hB.ctypes.data, cB.ctypes.data, eB.ctypes.data, kB.ctypes.data, pB.ctypes.data,
                                  C.byref(n_out), oh.ctypes.data, oc.ctypes.data, oe.ctypes.data, ok.ctypes.data, op.ctypes.data))
        n = n_out.value
        kc = np.zeros(n, dtype=KC_DTYPE)
        kc["hash"], kc["count"], kc["extra_count"] = oh[:n], oc[:n], oe[:n]
        km, pos, tk = ok[:n].copy(), op[:n].copy(), tk + tk2
    return kc, km, pos, tk


def merge_wire(params: SketchParams, bufs: List[np.ndarray], pad_n: int):
    """N packed partial sketches (pack_partial layout) -> merged (kc, km, pos, total_kmers) in one C call"""
    L = _lib.load()
    k = params.kmer_length
    kind = {"mash": 0, "scaled": 1}[params.kind]
    bufs = [np.ascontiguousarray(b, dtype=np.int64) for b in bufs]
    cap = int(sum(int(b[0]) for b in bufs))
    if params.kind == "mash":
        cap = min(cap, params.kmers_to_sketch)
    cap = max(cap, 1)
    oh, oc, oe = np.empty(cap, np.uint64), np.empty(cap, np.uint32), np.empty(cap, np.uint32)
    ok, op = np.empty((cap, k), np.uint8), np.empty(cap, np.uint64)
    ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
    n_out, tk = C.c_uint64(), C.c_uint64()
    check(L.fh_merge_wire(kind, params.kmers_to_sketch, params.scale, k, pad_n, len(bufs), ptrs, C.byref(n_out),
                          oh.ctypes.data, oc.ctypes.data, oe.ctypes.data, ok.ctypes.data, op.ctypes.data, C.byref(tk)))
    n = n_out.value
    kc = np.empty(n, dtype=KC_DTYPE)
    kc["hash"], kc["count"], kc["extra_count"] = oh[:n], oc[:n], oe[:n]
    return kc, ok[:n], op[:n], int(tk.value)


def sketch_device_blocks(sketchers, blocks, lens, stream_offsets) -> None:
    """fh_sketch_device_blocks: block i (a device pointer on sketcher i's device) through sketcher i, each on a library thread
    of its own, partial sketches merged into sketchers[0] -- ONE call, no Python in the per-device loop.  Afterwards
    sketchers[0].to_arrays() / .finish() deliver the merged sketch."""
    L = _lib.load()
    n = len(sketchers)
    if not (n == len(blocks) == len(lens) == len(stream_offsets)) or n == 0:
        raise ValueError("one block, length and stream offset per sketcher")
    hs = (C.c_void_p * n)(*[s._h for s in sketchers])
    bs = (C.c_void_p * n)(*[int(b) for b in blocks])
    ls = (C.c_uint64 * n)(*[int(x) for x in lens])
    os_ = (C.c_uint64 * n)(*[int(x) for x in stream_offsets])
    check(L.fh_sketch_device_blocks(hs, bs, ls, os_, n))


def gather_and_merge(dist, params: SketchParams, partial: tuple, pad_n: int, device=None):
    """rank 0 returns the merged sketch, the other ranks None.  `dist` = torch.distributed (any backend)."""
    import torch
    k = params.kmer_length
    kc, km, pos, tk = partial
    t = torch.from_numpy(pack_partial(kc, km, pos, tk, pad_n, k))
    if device is not None:
        t = t.to(device)
    rank, world = dist.get_rank(), dist.get_world_size()
    outs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, outs, dst=0)
    if rank != 0:
        return None
    return merge_wire(params, [o.cpu().numpy() for o in outs], pad_n)
