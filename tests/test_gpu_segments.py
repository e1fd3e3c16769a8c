"""The segment form of the sketch kernel (fh_k2s.hip; fh_set_record_stride / the stride probe of large blocks): the sketch
must be the oracle's bit for bit whatever the stride says -- right, wrong, or on a stream that has no records of one length
at all.  canonical_kmers yields len - k + 1 windows per record (mash.rs:76): with the right stride the kernel hashes exactly
those and skips the k positions per record whose window crosses the breaker."""
import os
import subprocess
import sys

import numpy as np
import pytest

import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O
from test_gpu_parity import assert_same, random_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def packed(reads):
    return np.frombuffer(b"".join(r + b"\0" for r in reads), dtype=np.uint8)


def fixed_reads(rng, n, L, genome, p_n=0.004, p_lower=0.02):
    return random_reads(rng, n, L, L, p_n=p_n, p_lower=p_lower, genome=genome)


def sketch_on_device(params, stream, stride, pushes=1, max_launch=0):
    sk = params.create_sketcher(max_launch=max_launch)
    sk.set_record_stride(stride)
    bufs = []
    # (blocks are sketched independently, as the records of one: cut between records, at 16-byte aligned places)
    rec16 = 16 * (stride if stride > 1 else 151)
    cut = [0] + [len(stream) * i // pushes // rec16 * rec16 for i in range(1, pushes)] + [len(stream)]
    for a, b in zip(cut[:-1], cut[1:]):
        d = F.DeviceBuffer(b - a + 256)
        d.upload(stream[a:b])
        bufs.append(d)
        sk.push_device(d.ptr, b - a)
    sk.sync()
    return sk, bufs


def oracle_of(kind, n, k, stream, scale=0.0):
    ora = O.OracleSketcher(kind, n, k, 0, scale) if kind == O.SCALED else O.OracleSketcher(kind, n, k, 0)
    ora.process_packed(stream, 0)
    return ora


@pytest.fixture(scope="module")
def genome():
    return S.synth_genome_host(300_000, 77)


# (long rounds: K <= 24 and 26; rounds of 16 in pairs: 25, 27..32 -- fh_core.h seg_long; 1 and 16: the 48-position cap, no pre-shift)
@pytest.mark.parametrize("k", [1, 5, 12, 16, 17, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32])
def test_right_stride_every_kernel_shape(genome, k):
    rng = np.random.default_rng(1000 + k)
    stream = packed(fixed_reads(rng, 3000 + k, 150, genome))  # (not a multiple of 64 records: a partial last tile)
    sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, 0), stream, 151)
    assert sk.debug_segments()[0] > 0 and sk.debug_segments()[2] == 151
    assert_same(sk, oracle_of(O.MASH, 1000, k, stream), "k=%d" % k)


@pytest.mark.parametrize("k", [33, 40, 47, 53, 64])
def test_two_word_kmers(genome, k):
    """fh_k2ws.hip: right stride, wrong stride, ragged records, a seed, a scaled sketch that stops waves inside tiles"""
    rng = np.random.default_rng(5000 + k)
    stream = packed(fixed_reads(rng, 3000 + k, 150, genome))
    for stride in (151, 100, 168):
        sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, 0), stream, stride)
        assert sk.debug_segments()[0] > 0
        assert_same(sk, oracle_of(O.MASH, 1000, k, stream), "k=%d stride=%d" % (k, stride))
    ragged = packed(random_reads(rng, 3000, 0, 220, p_n=0.01, genome=genome))
    sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, 0), ragged, 151)
    assert_same(sk, oracle_of(O.MASH, 1000, k, ragged), "ragged k=%d" % k)
    sk, _ = sketch_on_device(F.SketchParams.mash(500, 500, True, k, 42), stream, 151)
    ora = O.OracleSketcher(O.MASH, 500, k, 42)
    ora.process_packed(stream, 0)
    assert sk.debug_segments()[0] > 0
    assert_same(sk, ora, "seed 42 k=%d" % k)
    sk, _ = sketch_on_device(F.SketchParams.scaled(1000, k, 0.5, 0), stream, 151)
    assert_same(sk, oracle_of(O.SCALED, 1000, k, stream, 0.5), "scaled k=%d" % k)


@pytest.mark.parametrize("stride", [40, 97, 100, 150, 152, 167, 168])
@pytest.mark.parametrize("k", [21, 31])
def test_wrong_stride_changes_nothing(genome, k, stride):
    rng = np.random.default_rng(2000 + stride)
    stream = packed(fixed_reads(rng, 2500, 150, genome))
    sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, 0), stream, stride)
    assert sk.debug_segments()[0] > 0
    assert_same(sk, oracle_of(O.MASH, 1000, k, stream), "k=%d stride=%d" % (k, stride))


@pytest.mark.parametrize("L", [39, 60, 100, 125, 167])
def test_other_read_lengths(genome, L):
    rng = np.random.default_rng(3000 + L)
    stream = packed(fixed_reads(rng, 4000, L, genome))
    for k in (21, 31):
        sk, _ = sketch_on_device(F.SketchParams.mash(500, 500, True, k, 0), stream, L + 1)
        assert sk.debug_segments()[0] > 0
        assert_same(sk, oracle_of(O.MASH, 500, k, stream), "L=%d k=%d" % (L, k))


def test_streams_that_are_not_records_of_one_length(genome):
    rng = np.random.default_rng(4)
    ragged = packed(random_reads(rng, 4000, 0, 220, p_n=0.01, genome=genome))     # every length, empty records too
    one = np.concatenate([genome[:250_000], np.zeros(1, np.uint8)])               # one long record, no breaker inside
    ns = packed([bytes(r) for r in np.full((2000, 150), ord("N"), np.uint8)])     # nothing valid at all
    tail_n = packed([r[:140] + b"N" * 10 for r in fixed_reads(rng, 2000, 150, genome)])  # the last windows of every record invalid
    for name, stream in (("ragged", ragged), ("one", one), ("all N", ns), ("tail N", tail_n)):
        for k in (21, 31):
            sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, 0), stream, 151)
            assert_same(sk, oracle_of(O.MASH, 1000, k, stream), "%s k=%d" % (name, k))


def test_breaker_positions_holding_bases(genome):
    """the stride is right for most records, but some records are one base longer / shorter: the bytes where breakers
    'should' be are bases there, and the rounds that would be skipped hold k-mers"""
    rng = np.random.default_rng(5)
    reads = fixed_reads(rng, 3000, 150, genome)
    for i in range(100, 3000, 97):
        reads[i] = reads[i] + b"A" if i & 1 else reads[i][:-1]
    stream = packed(reads)
    for k in (21, 31):
        sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, 0), stream, 151)
        assert_same(sk, oracle_of(O.MASH, 1000, k, stream), "k=%d" % k)


@pytest.mark.parametrize("max_launch", [4096, 16384, 0])
@pytest.mark.parametrize("k", [21, 31, 48])
def test_loose_thresholds_stop_waves_inside_a_tile(genome, k, max_launch):
    """a scaled sketch that keeps half of all k-mers, on few waves (max_launch: 16 or 4 of them): every wave runs into its insert
    budget again and again, stops at the end of a round and hands the rest of its range back -- ONE leftover entry per wave
    (first tile, end of range, round to resume at), so that a relaunch, which has as many waves as entries, leaves none unread.
    (Round 5's first version wrote two entries per stopping wave; the second half of a list was never read: found by the fuzzer
    under option seg_stride=151, seed 525252 case 340.)"""
    rng = np.random.default_rng(6)
    stream = packed(random_reads(rng, 150, 0, 5000, p_n=0.001, genome=genome))
    for stride in (151, 100):
        sk, _ = sketch_on_device(F.SketchParams.scaled(1000, k, 0.5, 0), stream, stride, max_launch=max_launch)
        assert sk.debug_segments()[0] > 0
        if max_launch:
            assert sk.debug_counters()["relaunches"] > 0
        assert_same(sk, oracle_of(O.SCALED, 1000, k, stream, 0.5), "scaled 0.5 k=%d stride=%d max_launch=%d" % (k, stride, max_launch))


def test_oversketch_the_input_cannot_fill(genome):
    rng = np.random.default_rng(6)
    stream = packed(fixed_reads(rng, 6000, 150, genome, p_n=0.001))
    sk, _ = sketch_on_device(F.SketchParams.mash(400_000, 400_000, True, 31, 0), stream, 151)
    assert_same(sk, oracle_of(O.MASH, 400_000, 31, stream), "oversketch")


def test_low_complexity_fails_the_speculation(genome):
    rng = np.random.default_rng(7)
    few = fixed_reads(rng, 40, 150, genome, p_n=0.0)
    stream = packed([few[i % 40] for i in range(6000)])  # 40 reads, 150 times each: far fewer distinct hashes than positions
    for k in (21, 31):
        sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, 0), stream, 151)
        assert_same(sk, oracle_of(O.MASH, 1000, k, stream), "k=%d" % k)


def test_blocks_pushed_one_after_the_other(genome):
    rng = np.random.default_rng(8)
    stream = packed(fixed_reads(rng, 6400, 150, genome))
    sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, 21, 0), stream, 151, pushes=3)
    assert_same(sk, oracle_of(O.MASH, 1000, 21, stream), "three pushes")


def test_stride_found_by_the_probe():
    """a block of whole records of one length is recognised without being told, N's inside the first record included; a ragged
    block is not.  A handle's first block waits for the answer if it is large (option seg_probe_wait_min; 256 MiB by default);
    otherwise it is asked behind its own launches and the NEXT block of the handle goes by the answer (option seg_probe_min: from
    which size on; 16 MiB by default)"""
    code = r'''
import numpy as np, os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O
from test_gpu_parity import assert_same, random_reads
g = S.synth_genome_host(200000, 3)
rng = np.random.default_rng(9)
def run(sk, reads, want_stride):
    stream = np.frombuffer(b"".join(r + b"\0" for r in reads), dtype=np.uint8)
    d = F.DeviceBuffer(stream.size + 256); d.upload(stream)
    sk.reset(); l0 = sk.debug_segments()[0]
    sk.push_device(d.ptr, stream.size); sk.sync()
    launches, probes, stride = sk.debug_segments()
    assert probes >= 1 and stride == want_stride and (launches > l0) == (want_stride != 0), (launches, l0, probes, stride, want_stride)
    ora = O.OracleSketcher(O.MASH, 1000, 21, 0); ora.process_packed(stream, 0)
    assert_same(sk, ora)
reads = random_reads(rng, 3000, 150, 150, p_n=0.002, genome=g)
reads[0] = reads[0][:30] + b"NN" + reads[0][32:]
ragged = random_reads(rng, 3000, 100, 150, genome=g)
waits = F.get_option("seg_probe_wait_min") == "0"
sk = F.SketchParams.mash(1000, 1000, True, 21, 0).create_sketcher()
run(sk, reads, 151 if waits else 0)        # the handle's first block: by its own answer only if it waits for it
run(sk, reads, 151)                        # the next one goes by what the first said
run(sk, ragged, 151)                       # ... also when that is wrong for it (nothing but speed depends on it)
run(sk, ragged, 0)                         # and the ragged block's own answer is "none"
run(sk, random_reads(rng, 3000, 100, 100, genome=g), 0)
run(sk, random_reads(rng, 3000, 100, 100, genome=g), 101)
print("probe OK")
''' % (ROOT, ROOT)
    for wait_min in ("0", "1000000000000"):
        r = subprocess.run([sys.executable, "-c", code], env=F.debug_env(seg_probe_min="0", seg_probe_wait_min=wait_min),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0 and "probe OK" in r.stdout, (wait_min, r.stdout[-3000:])


@pytest.mark.parametrize("stride", [41, 44, 45, 46, 64, 65, 88, 89, 129, 133])
@pytest.mark.parametrize("k", [21, 24, 26, 27, 30])
def test_strides_around_the_round_lengths(genome, k, stride):
    """a round is 65 - K positions (K <= 24, 26) or a pair of 16: strides one short of, equal to and one past a whole number of
    rounds, with reads of exactly that length (the last round holds one window, none, or a full set) and a loose threshold that
    stops waves between rounds"""
    rng = np.random.default_rng(7000 + 100 * k + stride)
    stream = packed(fixed_reads(rng, 1500 + stride, stride - 1, genome))
    sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, 0), stream, stride)
    assert sk.debug_segments()[0] > 0 and sk.debug_segments()[2] == stride
    assert_same(sk, oracle_of(O.MASH, 1000, k, stream), "k=%d stride=%d" % (k, stride))
    sk, _ = sketch_on_device(F.SketchParams.scaled(100, k, 0.5, 0), stream, stride, max_launch=4096)
    assert_same(sk, oracle_of(O.SCALED, 100, k, stream, 0.5), "scaled k=%d stride=%d" % (k, stride))


# --- records longer than a lane's segment: two or four lanes to a record (SketchArgs::seg_sub; strides 169..672) ---
@pytest.mark.parametrize("L,k", [(168, 21), (200, 21), (250, 21), (250, 31), (300, 21), (300, 31), (335, 16), (336, 24), (400, 21), (500, 31),
                                  (671, 21), (671, 32), (250, 1), (301, 27)])
def test_long_records_two_and_four_lanes_each(L, k):
    genome = S.synth_genome_host(400_000, 78)
    rng = np.random.default_rng(6000 + L + k)
    stream = packed(fixed_reads(rng, 1500 + k, L, genome))  # (a partial last tile)
    sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, 0), stream, L + 1)
    assert sk.debug_segments()[0] > 0 and sk.debug_segments()[2] == L + 1
    assert_same(sk, oracle_of(O.MASH, 1000, k, stream), "L=%d k=%d" % (L, k))


@pytest.mark.parametrize("stride", [169, 251, 252, 300, 336, 337, 499, 672])
def test_long_strides_that_are_wrong_change_nothing(stride):
    """strides of the two- and four-lane forms on streams they do not fit: reads of 150 and of 250 bases, ragged records, one long
    record, breakers where bases should be -- the sketch never depends on what the stride says"""
    genome = S.synth_genome_host(300_000, 79)
    rng = np.random.default_rng(7000 + stride)
    streams = {
        "150s": packed(fixed_reads(rng, 2500, 150, genome)),
        "250s": packed(fixed_reads(rng, 1500, 250, genome)),
        "ragged": packed(random_reads(rng, 3000, 0, 420, p_n=0.01, genome=genome)),
        "one": np.concatenate([genome[:250_000], np.zeros(1, np.uint8)]),
    }
    for name, stream in streams.items():
        for k in (21, 31):
            sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, 0), stream, stride)
            assert sk.debug_segments()[0] > 0, (name, k)
            assert_same(sk, oracle_of(O.MASH, 1000, k, stream), "%s k=%d stride=%d" % (name, k, stride))


@pytest.mark.parametrize("k,seed", [(21, 42), (31, 2**63 + 11), (16, 1), (25, 7)])
def test_seeds_go_through_the_segment_kernels(genome, k, seed):
    """hash_seed != 0 (hashing.rs:10-12: murmurhash3_x64_128(kmer, seed)): the segment kernels take the seed like every other"""
    rng = np.random.default_rng(8000 + k)
    for L in (150, 250):
        stream = packed(fixed_reads(rng, 2500, L, genome))
        sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, seed), stream, L + 1)
        assert sk.debug_segments()[0] > 0
        ora = O.OracleSketcher(O.MASH, 1000, k, seed)
        ora.process_packed(stream, 0)
        assert_same(sk, ora, "L=%d k=%d seed=%d" % (L, k, seed))
    sc = F.SketchParams.scaled(1000, k, 0.25, seed)
    sk, _ = sketch_on_device(sc, stream, 251, max_launch=16384)  # few waves, loose threshold: waves stop inside tiles
    ora = O.OracleSketcher(O.SCALED, 1000, k, seed, 0.25)
    ora.process_packed(stream, 0)
    assert_same(sk, ora, "scaled k=%d seed=%d" % (k, seed))


def test_records_of_many_lengths_go_through_the_work_item_form():
    """SEG_RAGGED (fh_k2s.hip; k = 25, 27..32): no stride fits, the lanes' cells of 32 positions become work items ordered by size
    and dealt out 64 a round.  Forced on every block without a stride (option seg_ragged=1, read once per process: a child), on
    ragged reads, trimmed mixes, one long record, all-N, tails of N, few waves with loose thresholds (waves stop inside tiles and
    resume at a round of a rebuilt item list), seeds -- the sketch is the oracle's"""
    code = r'''
import numpy as np
import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O
import sys
sys.path.insert(0, "tests")
from test_gpu_parity import assert_same, random_reads
from test_gpu_segments import packed, fixed_reads, sketch_on_device, oracle_of
genome = S.synth_genome_host(300_000, 77)
rng = np.random.default_rng(99)
streams = {
    "ragged 0..220": packed(random_reads(rng, 6000, 0, 220, p_n=0.01, genome=genome)),
    "trimmed 35..150": packed(random_reads(rng, 8000, 35, 150, p_n=0.002, genome=genome)),
    "mostly 150": packed([r if i % 5 else r[:int(rng.integers(30, 150))] for i, r in enumerate(fixed_reads(rng, 6000, 150, genome))]),
    "one record": np.concatenate([genome[:250_000], np.zeros(1, np.uint8)]),
    "all N": packed([bytes(r) for r in np.full((3000, 150), ord("N"), np.uint8)]),
    "tail N": packed([r[:120] + b"N" * 30 for r in fixed_reads(rng, 3000, 150, genome)]),
}
n_rag = 0
for k in (25, 27, 28, 29, 30, 31, 32):
    for name, stream in streams.items():
        if k not in (25, 31) and name not in ("ragged 0..220", "trimmed 35..150"):
            continue
        sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, k, 0), stream, 0)
        launches, _, stride = sk.debug_segments()
        assert launches > 0 and stride == 128, (k, name, launches, stride)
        n_rag += 1
        assert_same(sk, oracle_of(O.MASH, 1000, k, stream), "%s k=%d" % (name, k))
# the K whose rounds are longer than 32 positions keep the tile kernel
sk, _ = sketch_on_device(F.SketchParams.mash(1000, 1000, True, 21, 0), streams["trimmed 35..150"], 0)
assert sk.debug_segments()[0] == 0
assert_same(sk, oracle_of(O.MASH, 1000, 21, streams["trimmed 35..150"]), "k=21")
# seeds, a scaled sketch that keeps half of all k-mers on few waves
stream = streams["trimmed 35..150"]
for k, seed in ((31, 42), (27, 2**63 + 9)):
    sk, _ = sketch_on_device(F.SketchParams.mash(500, 500, True, k, seed), stream, 0)
    ora = O.OracleSketcher(O.MASH, 500, k, seed); ora.process_packed(stream, 0)
    assert sk.debug_segments()[2] == 128
    assert_same(sk, ora, "seed k=%d" % k)
for ml in (4096, 16384, 0):
    sk, _ = sketch_on_device(F.SketchParams.scaled(1000, 31, 0.5, 0), stream, 0, max_launch=ml)
    assert sk.debug_segments()[2] == 128
    assert_same(sk, oracle_of(O.SCALED, 1000, 31, stream, 0.5), "scaled max_launch=%d" % ml)
print("child ok", n_rag)
'''
    r = subprocess.run([sys.executable, "-c", code], env=F.debug_env(seg_ragged="1"), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-3000:]
