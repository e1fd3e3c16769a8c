"""The host-side FASTQ strip (csrc/fh_fqstrip.h): plain 4-line FASTQ text -> the packed sequence stream by a team of threads.
CPU only.  Held against a line-by-line restatement in Python for every thread count, line ending and chunk shape; text that is
not plain 4-line FASTQ must be refused (the caller then lets the host parser -- the judge of what needletail accepts -- read it)."""
import numpy as np
import pytest

from finch_rs_amd import host as H
from finch_rs_amd.sketch_schemes import FinchError


def fastq(rng, n, lo=0, hi=300, eol=b"\n", last_eol=True, long_headers=False, blanks=False):
    recs = []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTNacgt", np.uint8), size=L))
        if blanks and L > 4 and i % 7 == 0:
            seq = seq[:2] + b" " + seq[2:-1] + b"\t" + seq[-1:]
        qual = bytes(rng.integers(33, 74, size=len(seq), dtype=np.uint8))
        if i % 5 == 0:
            qual = b"@" + qual[1:] if qual else qual  # quality lines that begin with '@' (and '+') are what makes FASTQ hard to split
        if i % 11 == 0 and qual:
            qual = b"+" + qual[1:]
        hdr = b"@r%d" % i + (b" some:longer:description/1 %d" % (i * 7919) if long_headers else b"")
        recs.append(hdr + eol + seq + eol + b"+" + (hdr[1:] if i % 3 == 0 else b"") + eol + qual)
    return eol.join(recs) + (eol if last_eol else b"")


def reference(text, eol, last_eol):
    lines = text.split(b"\n")
    if last_eol:
        assert lines[-1] == b""
        lines.pop()
    elif eol == b"\r\n":
        pass  # (the last line simply has no line end: nothing to trim from it)
    assert len(lines) % 4 == 0
    out, bases = [], 0
    for i in range(0, len(lines), 4):
        seq = lines[i + 1].rstrip(b"\r") if eol == b"\r\n" else lines[i + 1]
        bases += len(seq)
        kept = bytes(c for c in seq if c not in b" \t\r\n")
        out.append(kept + b"\0" * (len(seq) - len(kept)) + b"\0")
    return b"".join(out), len(lines) // 4, bases


@pytest.mark.parametrize("threads", [1, 2, 3, 7, 16, 33])
@pytest.mark.parametrize("eol,last_eol", [(b"\n", True), (b"\n", False), (b"\r\n", True), (b"\r\n", False)])
def test_strip_matches_the_line_by_line_restatement(threads, eol, last_eol):
    rng = np.random.default_rng(threads * 10 + len(eol) + last_eol)
    for n, lo, hi, lh, bl in ((1, 5, 5, False, False), (3, 0, 0, False, False), (400, 0, 300, True, False), (1500, 150, 150, False, True),
                              (60, 1000, 3000, False, False)):
        text = fastq(rng, n, lo, hi, eol, last_eol, lh, bl)
        packed, nrec, bases = H.fastq_strip_probe(text, threads)
        want, wrec, wbases = reference(text, eol, last_eol)
        assert (nrec, bases) == (wrec, wbases)
        # (the threads' regions are joined by breaker bytes: the same records in the same order, zeros between them)
        assert [x for x in packed.split(b"\0") if x] == [x for x in want.split(b"\0") if x], (threads, n, lo, hi)
        assert packed.count(b"\0") >= nrec and (not packed or packed[-1] == 0)


def test_what_is_not_plain_four_line_fastq_is_refused():
    rng = np.random.default_rng(5)
    good = fastq(rng, 50, 20, 80)
    lines = good.split(b"\n")
    bad = {
        "blank line between records": b"\n".join(lines[:8] + [b""] + lines[8:]),
        "sequence over two lines": b"\n".join(lines[:5] + [lines[5][:10], lines[5][10:]] + lines[6:]),
        "quality shorter than the sequence": b"\n".join(lines[:7] + [lines[7][:-1]] + lines[8:]),
        "header without '@'": b"\n".join([b"r0"] + lines[1:]),
        "separator without '+'": b"\n".join(lines[:2] + [b"-"] + lines[3:]),
        "cut off inside a record": b"\n".join(lines[:-3]),
        "two trailing newlines": good + b"\n",
        "FASTA": b">x\nACGT\n",
    }
    for name, text in bad.items():
        for threads in (1, 4, 9):
            with pytest.raises(FinchError, match="not plain 4-line FASTQ"):
                H.fastq_strip_probe(text, threads)
    # and the empty text is zero records
    assert H.fastq_strip_probe(b"", 4) == (b"", 0, 0)


def test_a_wrong_guess_of_a_record_start_fails_the_chunk_instead_of_the_sketch():
    """quality lines made to look like records: '@...', then a line, then '+...', then a line, then '@' -- a thread whose stretch
    begins there guesses wrong; the walk of the thread in front does not land on the guess and the chunk is refused (the caller
    then reads the input through the other paths), or the guess is never reached -- never a wrong stream"""
    recs = []
    for i in range(3000):
        seq = b"+" + b"ACGT" * 10 if i % 2 else b"ACGTACGTAC" * 4 + b"A"
        qual = (b"@" + b"I" * (len(seq) - 1))
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, seq, qual))
    text = b"".join(recs)
    lines = text.split(b"\n")[:-1]
    want = b"".join(lines[i + 1] + b"\0" for i in range(0, len(lines), 4))
    for threads in (1, 2, 5, 16, 64):
        try:
            packed, nrec, bases = H.fastq_strip_probe(text, threads)
        except FinchError:
            continue
        assert nrec == 3000 and [x for x in packed.split(b"\0") if x] == [x for x in want.split(b"\0") if x]
