"""The reported counts are u32 and SATURATE (mash.rs:45-50: `count.0 = count.0.saturating_add(1)`, `count.1` likewise): the
device keeps 64-bit counters per strand and clamps at to_vec and in the merges.  2^32 occurrences of a k-mer are out of a
test's reach, so a test hook (fh_debug_add_counts) moves the counters of everything held so far close to the limit; the
oracle's counts on the same occurrences say what the unsaturated totals are."""
import numpy as np
import pytest

import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu
U32 = 2**32 - 1


def _expected(stream, n, k, add_count, add_extra):
    """the oracle on the stream twice (the occurrences before and after the hook), its counts as exact integers, then the
    hook's additions and the reference's saturation"""
    ora = O.OracleSketcher(O.MASH, n, k, 0)
    ora.process_packed(stream, 0)
    ora.process_packed(stream, 0)
    kc, km = ora.to_vec()
    count = kc["count"].astype(np.uint64)
    extra = kc["extra_count"].astype(np.uint64)
    assert int(count.max()) < 10**6  # (the oracle itself is nowhere near saturating here)
    # count (mash.rs:46) counts every occurrence, extra_count (mash.rs:47) the reverse-strand ones: the hook adds add_count to
    # the forward-strand counter and add_extra to the reverse-strand one
    return kc["hash"], np.minimum(count + np.uint64(add_count + add_extra), U32), np.minimum(extra + np.uint64(add_extra), U32), km


@pytest.mark.parametrize("k,n", [(21, 1000), (31, 300), (40, 200)])
@pytest.mark.parametrize("add_count,add_extra", [(U32 - 2, 0), (0, U32 - 1), (U32 - 3, U32 - 3), (2**33, 5)])
def test_counts_saturate_like_the_reference(k, n, add_count, add_extra):
    g = S.synth_genome_host(60_000, 11)
    stream = S.synth_reads_host(g, 0, 4000, 150, 5, 10000, 500)  # ~10-fold coverage: counts of 1..30, both strands
    sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher()
    sk.push_block(stream)
    sk.debug_add_counts(add_count, add_extra)
    sk.push_block(stream)
    kc, km, _ = sk.to_arrays()
    h, c, e, okm = _expected(stream, n, k, add_count, add_extra)
    assert np.array_equal(kc["hash"], h)
    assert np.array_equal(kc["count"].astype(np.uint64), c), (kc["count"][:8], c[:8])
    assert np.array_equal(kc["extra_count"].astype(np.uint64), e)
    assert np.array_equal(km, okm)
    assert (c == U32).any() or add_count + add_extra < U32 - 30  # the case really reaches the limit


def test_merge_of_partial_sketches_saturates():
    """partial sketches are merged with counts summed in u64 and clamped (== saturating adds)"""
    g = S.synth_genome_host(60_000, 12)
    stream = S.synth_reads_host(g, 0, 4000, 150, 6, 10000, 500)
    a = F.SketchParams.mash(500, 500, True, 21, 0).create_sketcher()
    b = F.SketchParams.mash(500, 500, True, 21, 0).create_sketcher()
    for sk, add in ((a, U32 - 40), (b, 30)):
        sk.push_block(stream)
        sk.debug_add_counts(add, add)
        sk.finish()
    a.merge(b)
    kc, km, _ = a.to_arrays()
    ora = O.OracleSketcher(O.MASH, 500, 21, 0)
    ora.process_packed(stream, 0)
    okc, okm = ora.to_vec()
    tot = 2 * okc["count"].astype(np.uint64) + np.uint64(2 * (U32 - 40) + 2 * 30)
    ext = 2 * okc["extra_count"].astype(np.uint64) + np.uint64(U32 - 40 + 30)
    assert np.array_equal(kc["hash"], okc["hash"])
    assert np.array_equal(kc["count"].astype(np.uint64), np.minimum(tot, U32))
    assert np.array_equal(kc["extra_count"].astype(np.uint64), np.minimum(ext, U32))
    assert (kc["count"] == U32).all() and (np.minimum(ext, U32) == U32).any()
