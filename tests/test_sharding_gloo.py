"""N>1 path on CPU: two processes over gloo shard one read set by read blocks, gather their partial
sketches to rank 0 and merge on the host (finch_rs_amd/sharding.py + fh_merge_partials).  There is no
GPU here, so each rank's sketcher is the oracle standing in for the device engine; what is under test
is the sharding arithmetic, the wire format and the merge, against the oracle run on the whole input."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["FH_ROOT"])
from finch_rs_amd import sharding as SH
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
kind = os.environ["FH_KIND"]
n, k, nr, rl, seed = 300, 21, 6001, 100, 5
params = S.SketchParams.mash(n, n, True, k, 0) if kind == "mash" else S.SketchParams.scaled(n, k, 0.002, 0)
g = S.synth_genome_host(50000, seed)
lo, hi = SH.shard_bounds(nr, rank, world)
reads = S.synth_reads_host(g, lo, hi - lo, rl, seed, 10000, 500)
o = O.OracleSketcher(O.MASH if kind == "mash" else O.SCALED, n, k, 0, 0.002)
o.process_packed(reads, 0)
kc, km = o.to_vec()
# the oracle has no positions: any value consistent with stream order works for distinct k-mers
pos = np.full(len(kc), lo * (rl + 1), dtype=np.uint64)
pad = 4096
merged = SH.gather_and_merge(dist, params, (kc, km, pos, o.total_bases_and_kmers()[1]), pad)
if rank == 0:
    whole = S.synth_reads_host(g, 0, nr, rl, seed, 10000, 500)
    w = O.OracleSketcher(O.MASH if kind == "mash" else O.SCALED, n, k, 0, 0.002)
    w.process_packed(whole, 0)
    wkc, wkm = w.to_vec()
    assert np.array_equal(merged[0], wkc), "hash/count mismatch"
    assert np.array_equal(merged[1], wkm), "kmer mismatch"
    assert merged[3] == w.total_bases_and_kmers()[1]
    print("MERGE_OK", len(wkc))
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("kind", ["mash", "scaled"])
@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_shard_gather_merge(kind, world):
    import __graft_entry__ as G
    G.build()
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FH_ROOT=ROOT, FH_KIND=kind)
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "MERGE_OK" in outs[0]


def test_shard_bounds_cover_everything():
    from finch_rs_amd import sharding as SH
    for n in [0, 1, 7, 8, 333333334]:
        for w in [1, 2, 3, 8]:
            b = [SH.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_pack_unpack_roundtrip():
    from finch_rs_amd import sharding as SH
    from finch_rs_amd.sketch_schemes import KC_DTYPE
    rng = np.random.default_rng(1)
    for k in [5, 21, 31, 32]:
        n = 37
        kc = np.zeros(n, dtype=KC_DTYPE)
        kc["hash"] = np.sort(rng.integers(0, 2**63, n).astype(np.uint64)) * np.uint64(2) + np.uint64(1)
        kc["count"] = rng.integers(1, 2**32 - 1, n)
        kc["extra_count"] = rng.integers(0, 2**31, n)
        km = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, k))
        pos = rng.integers(0, 2**62, n).astype(np.uint64)
        p = SH.pack_partial(kc, km, pos, 123456789012, 64, k)
        kc2, km2, pos2, tk = SH.unpack_partial(p, 64, k)
        assert np.array_equal(kc, kc2) and np.array_equal(km, km2) and np.array_equal(pos, pos2) and tk == 123456789012


@pytest.mark.parametrize("kind", ["mash", "scaled"])
def test_kway_merge_equals_the_chain_of_pairwise_merges(kind):
    """merge_wire (fh_merge_wire: one k-way pass over the packed partial sketches, what gather_and_merge uses) ==
    merge_partials (pairwise fh_merge_partials): shared hashes, counts that saturate, equal hashes with different
    k-mers (the smaller first position wins), empty partials, fewer than n hashes in total"""
    from finch_rs_amd import sharding as SH
    from finch_rs_amd.sketch_schemes import KC_DTYPE, SketchParams
    rng = np.random.default_rng(99)
    k = 21
    for trial in range(12):
        world = int(rng.integers(3, 9))
        n = int(rng.choice([5, 200, 1000]))
        params = SketchParams.mash(n, n, True, k, 0) if kind == "mash" else SketchParams.scaled(n, k, float(rng.choice([0.5, 0.01])), 0)
        pool_h = np.unique(rng.integers(0, 2**64 - 1, 3 * n + 7, dtype=np.uint64) >> np.uint64(int(rng.choice([0, 4, 40]))))
        pool_km = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(len(pool_h), k))
        parts = []
        for r in range(world):
            m = int(rng.integers(0, min(n, len(pool_h)) + 1)) if trial % 4 else 0 if r == 0 else min(n, len(pool_h))
            idx = np.sort(rng.choice(len(pool_h), size=m, replace=False))
            kc = np.zeros(m, dtype=KC_DTYPE)
            kc["hash"] = pool_h[idx]
            kc["count"] = rng.choice(np.array([1, 2, 7, 2**31, 2**32 - 1], dtype=np.uint64), size=m).astype(np.uint32)
            kc["extra_count"] = (kc["count"] // rng.integers(1, 4, m)).astype(np.uint32)
            km = pool_km[idx].copy()
            flip = rng.random(m) < 0.1          # a different k-mer under the same hash (64-bit collision across shards)
            km[flip] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(int(flip.sum()), k))
            pos = rng.permutation(10_000_000)[:m].astype(np.uint64) + np.uint64(r * 10_000_000)
            parts.append((kc, km, pos, int(rng.integers(0, 2**40))))
        a = SH.merge_partials(params, parts)
        pad = max(len(p[0]) for p in parts) + int(rng.integers(0, 5))
        b = SH.merge_wire(params, [SH.pack_partial(p[0], p[1], p[2], p[3], pad, k) for p in parts], pad)
        assert a[3] == b[3]
        assert np.array_equal(a[0], b[0]), (trial, kind)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (trial, kind)


def test_bench_merge_pipe_keeps_order_and_reraises():
    """bench.py's MergePipe: items are merged in the order they were put, flush() waits for all of them and hands back the last
    result, an exception in the merge surfaces on the caller's thread, and close() ends the worker"""
    import importlib.util
    import threading
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen, inits = [], []

    def fn(x):
        time.sleep(0.01)
        seen.append((x, threading.current_thread().name))
        return x * 10
    p = bench.MergePipe(fn, init=lambda: inits.append(threading.current_thread().name))
    assert p.flush() is None  # nothing handed over yet
    for i in range(7):
        p.put(i)  # (blocks when two are in flight)
    assert p.flush() == 60 and [s[0] for s in seen] == list(range(7))
    assert len(inits) == 1 and all(s[1] == inits[0] for s in seen) and inits[0] != threading.current_thread().name
    p.close()
    assert not p.t.is_alive()

    def boom(x):
        raise ValueError("merge failed on %d" % x)
    q = bench.MergePipe(boom)
    q.put(3)
    with pytest.raises(ValueError, match="merge failed on 3"):
        q.flush()
    q.close()
