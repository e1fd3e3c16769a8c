"""fh_text_prefetch (include/finch_hip.h): the reader's early start of a staging buffer's host-to-device copy must change
nothing but timing -- whatever the reader does with it (matching length, wrong length, a prefetch nobody consumes, a reset in
between), the sketch is the oracle's."""
import ctypes as C

import numpy as np
import pytest

import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _fastq(seed, n_reads, rl=120, genome=80_000):
    rng = np.random.default_rng(seed)
    g = S.synth_genome_host(genome, seed)
    recs = []
    for i in range(n_reads):
        st = int(rng.integers(0, len(g) - rl))
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, g[st:st + rl].tobytes(), b"I" * rl))
    return recs


def _oracle(recs, k, n):
    o = O.OracleSketcher(O.MASH, n, k, 0)
    assert o.sketch_stream(b"".join(recs)) == 2  # (the format it found: FASTQ)
    return o.to_vec() + (o.total_bases_and_kmers()[1],)


@pytest.mark.parametrize("mode", ["none", "match", "wrong_len", "other_slot_unused", "reset_between"])
def test_prefetch_changes_nothing_but_timing(mode):
    k, n = 21, 500
    recs = _fastq(11, 3000)
    chunks = [b"".join(recs[i:i + 500]) for i in range(0, len(recs), 500)]  # six chunks of whole records
    want = _oracle(recs, k, n)
    sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher(stage_bytes=1 << 20)
    L, h = sk._L, sk._h
    bufs = (C.c_void_p * 2)()
    cap, nxt = C.c_uint64(), C.c_int()

    def buffers():
        S.check(L.fh_text_buffers(h, bufs, C.byref(cap), C.byref(nxt)))
        return nxt.value

    def fill(slot, data):
        assert len(data) <= cap.value
        C.memmove(bufs[slot], data, len(data))

    if mode == "reset_between":
        # a stream that is abandoned after a prefetch: its text must never reach the next stream's sketch
        slot = buffers()
        junk = b"".join(_fastq(99, 400))
        fill(slot, junk)
        S.check(L.fh_text_prefetch(h, slot, len(junk)))
        sk.reset()
    slot = buffers()
    for i, ch in enumerate(chunks):
        fill(slot, ch)
        if mode in ("match", "reset_between"):
            S.check(L.fh_text_prefetch(h, slot, len(ch)))
        elif mode == "wrong_len":
            S.check(L.fh_text_prefetch(h, slot, len(ch) - 7))  # the push copies the whole chunk itself
        elif mode == "other_slot_unused" and i == 0:
            S.check(L.fh_text_prefetch(h, slot ^ 1, 4096))     # stale bytes of the other buffer: overwritten by chunk 1's push
        S.check(L.fh_push_fastq_text(h, len(ch)))
        slot ^= 1
    kc, km, _ = sk.to_arrays()
    assert np.array_equal(kc, want[0]) and np.array_equal(km, want[1]) and sk.finish()[1] == want[2]
    # bad arguments are refused, harmless ones ignored
    assert L.fh_text_prefetch(h, 2, 10) != 0 and L.fh_text_prefetch(h, -1, 10) != 0
    assert L.fh_text_prefetch(h, 0, 0) == 0 and L.fh_text_prefetch(h, 0, 1 << 40) == 0


def test_to_arrays_refills_the_callers_buffers():
    """HipSketcher.to_arrays(out=...): the arrays of an earlier call are filled again (views of them come back) when they have
    room, fresh ones are made when they do not; large sketches come out of the lazy copy-out path the same"""
    recs = _fastq(5, 8000, genome=600_000)  # ~500 k distinct 21-mers
    text = b"".join(recs)
    for n in (300, 200_000):  # the second leaves its wide columns on the device until asked (>= 128 k records)
        o = O.OracleSketcher(O.MASH, n, 21, 0)
        assert o.sketch_stream(text) == 2
        okc, okm = o.to_vec()
        sk = F.SketchParams.mash(n, n, True, 21, 0).create_sketcher()
        for r in recs:
            sk.process(r.split(b"\n")[1])
        a = sk.to_arrays()
        assert len(okc) == n and np.array_equal(a[0], okc) and np.array_equal(a[1], okm)
        keep = (a[0].copy(), a[1].copy())
        a[0]["hash"][:] = 0
        a[1][:] = 0
        b = sk.to_arrays(out=a)
        assert b[0].base is a[0] or b[0] is a[0] or np.shares_memory(b[0], a[0])
        assert np.array_equal(b[0], keep[0]) and np.array_equal(b[1], keep[1])
        small = (np.empty(1, a[0].dtype), np.empty((1, 21), np.uint8), np.empty(1, np.uint64))
        c = sk.to_arrays(out=small) if len(okc) > 1 else b
        assert np.array_equal(c[0], keep[0])
        sk.close()
