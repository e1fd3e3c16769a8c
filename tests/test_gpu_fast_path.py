"""The single-synchronisation path of small sketches (kmers_to_sketch <= 3000; fh_api.hip "fast"): a speculative first range
whose verdict is taken on the device, the rest of the input queued behind it gated on that verdict, the threshold refreshed
inside the launch from the histogram of new hashes, and fh_finish as one fused launch -- against the CPU oracle, bit for bit,
including the cases where what was queued does NOT go as planned (speculation fails: the step-by-step path takes over), and
with the knobs that switch the pieces off.  Plus fh_sketch_device_blocks: N resident read blocks -> one merged sketch in one
library call (the N-GPU driver of configs[3]; here N handles on the one GPU).  Needs a real MI355X: run with `-m gpu`."""
import os

import numpy as np
import pytest

import finch_rs_amd as F
from finch_rs_amd import sharding as SH
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SEED = 77
REC = 151


def _reads(n_reads, genome_len=2_000_000, sub_ppm=10_000, n_ppm=500, first=0, seed=SEED):
    g = S.synth_genome_host(genome_len, seed)
    return S.synth_reads_host(g, first, n_reads, 150, seed, sub_ppm, n_ppm)


def _oracle(data, n=1000, k=21, first_pos=0):
    ora = O.OracleSketcher(O.MASH, n, k, 0)
    ora.process_packed(data, first_pos)
    return ora


def _same(kc, km, tk, ora, ctx=""):
    okc, okm = ora.to_vec()
    assert len(kc) == len(okc), (ctx, len(kc), len(okc))
    for f in ("hash", "count", "extra_count"):
        assert np.array_equal(kc[f], okc[f]), (ctx, f)
    assert np.array_equal(km, okm), ctx
    assert tk == ora.total_bases_and_kmers()[1], ctx


@pytest.fixture(scope="module")
def big_block():
    """80 Mbases of reads: above the 64 M positions up to which a first block is speculated on as a whole, so the pass is
    speculative prefix (32 M positions) + verdict + gated main launch.  (A genome of 20 Mb: the guess assumes that at least a
    quarter of the prefix's k-mers are distinct; at 2 Mb -- the other tests here -- it sometimes holds and sometimes does
    not, and both ways must give the oracle's sketch.)"""
    data = _reads(530_000, genome_len=20_000_000)
    return data, _oracle(data)


def _device_pass(data, n=1000, k=21, **env):
    old = os.environ.get("FH_DEBUG")
    F.debug_set(**env)  # (options of the library: FH_DEBUG, include/finch_hip.h)
    try:
        F.load().fh_release_cached()  # (a parked handle would keep the previous setting of the creation-time knobs)
        buf = F.DeviceBuffer(len(data) + 64)
        buf.upload(data)
        sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher()
        out = []
        for _ in range(2):  # the second pass runs on a reset handle (fused reset kernel)
            sk.reset()
            sk.push_device(buf.ptr, len(data))
            kc, km, _ = sk.to_arrays()
            out.append((kc.copy(), km.copy(), sk.finish()[1], sk.debug_counters()))
        sk.close()
        return out
    finally:
        if old is None:
            os.environ.pop("FH_DEBUG", None)
        else:
            os.environ["FH_DEBUG"] = old
        F.load().fh_release_cached()


def test_prefix_verdict_and_gated_main_launch_in_one_synchronisation(big_block):
    data, ora = big_block
    for i, (kc, km, tk, dbg) in enumerate(_device_pass(data)):
        _same(kc, km, tk, ora, "pass %d" % i)
        # one speculative range per pass, never looked at before fh_finish, which was the one fused launch; the launch over
        # the rest ran once (threshold refreshed inside it: no stop / prune / relaunch cycle)
        assert dbg["spec_deferred"] == i + 1 and dbg["spec_recovered"] == 0 and dbg["fused_finishes"] == i + 1, dbg
        assert dbg["relaunches"] == 0 and dbg["launches"] == 2 * (i + 1), dbg


@pytest.mark.parametrize("env", [{"no_fast": 1}, {"no_hist": 1}, {"no_spec": 1}, {"unit_tiles": 2}, {"max_range": 3_000_000}],
                         ids=lambda e: "+".join(e))
def test_knobs_that_switch_pieces_off_give_the_same_sketch(big_block, env):
    data, ora = big_block
    for kc, km, tk, dbg in _device_pass(data, **env):
        _same(kc, km, tk, ora, str(env))
        if "no_fast" in env:
            assert dbg["spec_deferred"] == 0 and dbg["fused_finishes"] == 0


@pytest.mark.parametrize("n,k", [(1, 21), (10, 5), (1000, 31), (3000, 21), (1000, 33), (500, 64)])
def test_other_sizes_and_kmer_lengths(n, k):
    data = _reads(470_000, seed=5)  # 70.5 M positions: prefix + gated main launch
    for kc, km, tk, dbg in _device_pass(data, n, k):
        _same(kc, km, tk, _oracle(data, n, k), "n=%d k=%d" % (n, k))


def test_failed_speculation_is_finished_step_by_step():
    """a block whose k-mers are few: the guessed threshold (4 x size hashes expected if every k-mer were new) leaves fewer than
    `size` of them -- the verdict on the device is negative, the gated main launch does nothing, and the host recovers through
    the re-read for the hashes above the guess"""
    g = S.synth_genome_host(3000, 9)  # ~6000 distinct canonical 21-mers, every one of them thousands of times
    data = S.synth_reads_host(g, 0, 470_000, 150, 9, 0, 0)
    ora = _oracle(data)
    for kc, km, tk, dbg in _device_pass(data):
        _same(kc, km, tk, ora)
        assert dbg["spec_recovered"] >= 1 and dbg["spec_second_pass"] >= 1, dbg
    # ... and a whole-block speculation (<= 64 M positions) that fails with nothing queued behind it
    small = data[: 200_000 * REC]
    for kc, km, tk, dbg in _device_pass(small):
        _same(kc, km, tk, _oracle(small))
        assert dbg["spec_recovered"] >= 1, dbg


@pytest.mark.parametrize("genome_len,n_reads,env", [(500_000, 200_000, {}), (500_000, 470_000, {}), (120_000, 200_000, {}),
                                                     (500_000, 200_000, {"no_fast": 1}), (500_000, 200_000, {"no_spec_rescale": 1})],
                         ids=["whole block", "prefix", "few hashes below the guess", "undeferred", "old way"])
def test_speculation_that_falls_short_on_deep_coverage_is_reread_up_to_a_rescaled_threshold(genome_len, n_reads, env):
    """reads without errors at 60- to 250-fold coverage: the guess (every k-mer distinct) leaves some tens of the 1000 hashes; the
    range is read again up to a threshold scaled by how far short the count fell -- not, as until round 4, for every hash above
    the guess -- and the sketch is the oracle's either way"""
    g = S.synth_genome_host(genome_len, 21)
    data = S.synth_reads_host(g, 0, n_reads, 150, 21, 0, 0)
    ora = _oracle(data)
    for kc, km, tk, dbg in _device_pass(data, **env):
        _same(kc, km, tk, ora, str(dbg))
        assert dbg["spec_second_pass"] >= 1, dbg


def test_pushes_after_a_deferred_speculation_resolve_it_first():
    """the verdict of a speculation is read at the next call that needs it: a second push, fh_sync, fh_text_bases"""
    data = _reads(300_000, seed=3)
    cut = 120_000 * REC
    sk = F.SketchParams.default().create_sketcher()
    buf = F.DeviceBuffer(len(data) + 64)
    buf.upload(data)
    sk.push_device(buf.ptr, cut)
    sk.sync()
    sk.set_stream_offset(cut)
    sk.push_device(buf.ptr + cut, len(data) - cut)
    kc, km, _ = sk.to_arrays()
    _same(kc, km, sk.finish()[1], _oracle(data))
    sk.close()


@pytest.mark.parametrize("n_blocks", [1, 2, 5, 8])
def test_sketch_device_blocks_equals_the_oracle_on_the_union(n_blocks):
    """fh_sketch_device_blocks: the read set cut into contiguous read blocks (shard_bounds), one handle each, all on this
    box's one GPU; the merged sketch in handles[0] is the oracle's sketch of the whole read set, twice in a row"""
    n_reads = 520_000  # 78 Mbases: a single block runs prefix + main launch, eight blocks whole-block speculations
    data = _reads(n_reads, seed=11)
    ora = _oracle(data)
    buf = F.DeviceBuffer(len(data) + 64)
    buf.upload(data)
    params = F.SketchParams.default()
    sks = [params.create_sketcher() for _ in range(n_blocks)]
    bounds = [SH.shard_bounds(n_reads, r, n_blocks) for r in range(n_blocks)]
    assert (bounds[0][0] * REC) % 16 == 0
    # (device blocks must be 16-byte aligned: copy each block to a buffer of its own, as every GPU of a node would hold it)
    blocks = []
    for lo, hi in bounds:
        b = F.DeviceBuffer((hi - lo) * REC + 64)
        b.upload(data[lo * REC:hi * REC])
        blocks.append(b)
    for it in range(2):
        SH.sketch_device_blocks(sks, [b.ptr for b in blocks], [(hi - lo) * REC for lo, hi in bounds], [lo * REC for lo, _ in bounds])
        kc, km, _ = sks[0].to_arrays()
        _same(kc, km, sks[0].finish()[1], ora, "%d blocks, call %d" % (n_blocks, it))
    for s in sks:
        s.close()


def test_sketch_device_blocks_rejects_what_it_cannot_merge():
    a = F.SketchParams.default().create_sketcher()
    b = F.SketchParams.mash(1000, 1000, True, 31, 0).create_sketcher()
    buf = F.DeviceBuffer(4096)
    with pytest.raises(F.FinchHipError, match="incompatible"):
        SH.sketch_device_blocks([a, b], [buf.ptr, buf.ptr], [1024, 1024], [0, 1024])
    with pytest.raises(F.FinchHipError, match="same handle"):
        SH.sketch_device_blocks([a, a], [buf.ptr, buf.ptr], [1024, 1024], [0, 1024])
    a.close()
    b.close()


def test_sketch_device_blocks_soak_500_calls_random_blocks_one_bad_block_mid_series():
    """the N-device driver before its first real eight-GPU run: eight handles, 500 calls in a row with blocks of random sizes --
    empty, one read, ends that are no multiple of anything -- every 25th call's merged sketch held against the oracle; in the
    middle of the series one call gets a block that cannot be sketched (a pointer that is not 16-byte aligned): the call names the
    block, the calls after it are as good as the ones before, and the caller's HIP device is the one it was after every return"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # (the runtime the library is linked against: already loaded)

    def current_device():
        d = ctypes.c_int(-1)
        assert hip.hipGetDevice(ctypes.byref(d)) == 0
        return d.value
    N, CALLS = 8, 500
    rng = np.random.default_rng(808)
    n_reads = 60_000
    data = _reads(n_reads, seed=12)
    pool = F.DeviceBuffer(len(data) + 4096)
    pool.upload(data)
    params = F.SketchParams.default()
    sks = [params.create_sketcher() for _ in range(N)]
    assert hip.hipSetDevice(0) == 0
    checked = 0
    for call in range(CALLS):
        # read ranges of random sizes out of the resident reads: block i = reads [lo_i, hi_i); record starts are 151 bytes apart, so a
        # block's start is 16-byte aligned only for lo_i a multiple of 16
        los, his = [], []
        for i in range(N):
            kind = rng.integers(0, 6)
            lo = int(rng.integers(0, n_reads // 16 - 1)) * 16
            if kind == 0:
                hi = lo                                             # an empty block
            elif kind == 1:
                hi = lo + 1                                         # one read
            elif kind == 2:
                hi = lo + int(rng.integers(2, 70))                  # less than a segment tile
            else:
                hi = min(n_reads, lo + int(rng.integers(100, 20_000)))
            los.append(lo)
            his.append(hi)
        ptrs = [pool.ptr + lo * REC for lo in los]
        lens = [(hi - lo) * REC for lo, hi in zip(los, his)]
        offs = list(np.cumsum([0] + lens[:-1]))
        if call == CALLS // 2:
            bad = int(rng.integers(1, N))
            ptrs[bad] += 8  # not 16-byte aligned: fh_push_device refuses it
            lens[bad] = max(lens[bad], REC)
            with pytest.raises(F.FinchHipError, match="block %d " % bad):
                SH.sketch_device_blocks(sks, ptrs, lens, offs)
            assert current_device() == 0
            continue
        SH.sketch_device_blocks(sks, ptrs, lens, offs)
        assert current_device() == 0
        if call % 25 == 0 or call == CALLS // 2 + 1:
            union = np.concatenate([data[lo * REC:hi * REC] for lo, hi in zip(los, his)] + [np.zeros(0, np.uint8)])
            ora = O.OracleSketcher(O.MASH, 1000, 21, 0)
            if len(union):
                ora.process_packed(union, 0)
            kc, km, _ = sks[0].to_arrays()
            _same(kc, km, sks[0].finish()[1], ora, "call %d" % call)
            checked += 1
    assert checked >= 20
    for s in sks:
        s.close()
