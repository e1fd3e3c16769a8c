"""Extended randomised comparison of the device-side record splitters (fh_text.hip) with the host parser and the oracle.
    python tools/fuzz_device_text.py [cases [seed]]       (on an MI355X; FH_STAGE_BYTES=4096 makes the cuts land everywhere)
FASTA: random FASTA-shaped text (arbitrary bytes, '>' anywhere, LF / CRLF, blank lines, ragged lines, empty records).
FASTQ: four-line records with damage sprinkled in (blank lines, CRLF, '@' / '+' starting quality lines, a missing last newline,
sequence / quality lengths that differ, truncated records): the default path (device, host parser as the fallback) must give
what the host parser alone gives -- the same sketch or an error -- and, where the oracle's parser accepts the text, the
oracle's sketch.
FUZZ_FILES=1: every text is also written as a plain file, a gzip file and a bgzip-style BGZF file and sketched through
sketch_files (device-side splitting, the inflate sources, BGZF members inflated on the device for FASTQ): each must give what
the text gave -- the same sketch, or an error.
FUZZ_SHARDED=1: every text also through sketch_stream_sharded over three handles on device 0 with chunks of 16-40 KB
(one input across several devices: record-aligned chunks dealt round-robin, the k-1 base halo across FASTA cuts)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F  # noqa: E402
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 77000


def run(data, p, mode):
    if mode is None:
        F.debug_set(device_parse=None)
    else:
        F.debug_set(device_parse=mode)
    try:
        return H.sketch_stream(data, "x", p, H.FilterParams(False)).sketch(0), None
    except Exception as e:  # noqa
        return None, str(e)


def same(a, b):
    return (np.array_equal(a.arrays[0], b.arrays[0]) and np.array_equal(a.arrays[1], b.arrays[1])
            and (a.seq_length, a.num_valid_kmers) == (b.seq_length, b.num_valid_kmers))


def vs_oracle(b, data, p):
    o = O.OracleSketcher(O.MASH, p.kmers_to_sketch, p.kmer_length, 0)
    if o.sketch_stream(data) <= 0:
        return False
    okc, okm = o.to_vec()
    assert np.array_equal(b.arrays[0], okc) and np.array_equal(b.arrays[1], okm)
    assert (b.seq_length, b.num_valid_kmers) == o.total_bases_and_kmers()
    return True


import gzip, struct, zlib
FILES = os.environ.get("FUZZ_FILES") is not None
n_files = 0


def bgzf(data, block=65280):
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        ch = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = co.compress(ch) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(ch), len(ch)))
    return b"".join(out)


SHARDED = os.environ.get("FUZZ_SHARDED") is not None
n_sharded = 0


def check_sharded(ref, data, p, case, rng):
    global n_sharded
    F.debug_set(device_parse=None)
    for chunk in (int(rng.integers(16000, 40000)), 0):  # (a FASTQ record -- up to 6 KB here -- has to fit a chunk)
        try:
            r = H.sketch_stream_sharded(data, "x", p, H.FilterParams(False), [0, 0, 0], chunk).sketch(0)
            err = None
        except Exception as e:  # noqa
            r, err = None, str(e)
        assert (r is None) == (ref is None), (case, chunk, "sharded refused: %s" % err if r is None else "sharded accepted")
        if r is not None:
            assert same(r, ref), (case, chunk)
        n_sharded += 1


def check_files(ref, data, p, case):
    """ref: the sketch of the text through sketch_stream (None: it was refused)"""
    global n_files
    F.debug_set(device_parse=None)
    for ext, img in ((".txt", data), (".gz", gzip.compress(data, 1)), (".bgzf.gz", bgzf(data))):
        path = "/dev/shm/fuzz_text_%d%s" % (os.getpid(), ext)
        with open(path, "wb") as f:
            f.write(img)
        try:
            r = H.sketch_files([path], p, H.FilterParams(False), n_threads=1).sketch(0)
        except Exception as e:  # noqa
            r = None
        finally:
            os.remove(path)
        assert (r is None) == (ref is None), (case, ext, "file refused" if r is None else "file accepted")
        if r is not None:
            assert same(r, ref), (case, ext)
        n_files += 1


alpha = np.frombuffer(b"ACGTACGTACGTACGTacgtNnuU>>- \t\r\xff*@+", dtype=np.uint8)
qual = np.frombuffer(bytes(range(33, 100)), dtype=np.uint8)
n_fa = n_fq = n_fq_err = n_or = 0
verbose = os.environ.get("FUZZ_VERBOSE") is not None
for case in range(n_cases):
    if verbose:
        print("case", case, flush=True)
    rng = np.random.default_rng(seed0 + case)
    k = int(rng.choice([3, 11, 16, 21, 31, 33, 48]))
    p = S.SketchParams.mash(50, 50, True, k, 0)
    if case % 2 == 0:
        lines = []
        for i in range(int(rng.choice([1, 5, 60, 800, 4000]))):
            r = rng.random()
            if i == 0 or r < 0.08:
                lines.append(b">" + bytes(rng.choice(alpha, size=int(rng.integers(0, 40)))))
            elif r < 0.12:
                lines.append(b"")
            else:
                lines.append(bytes(rng.choice(alpha, size=int(rng.choice([1, 7, 60, 70, 500, 6000])))))
        eol = [b"\n", b"\r\n"][int(rng.integers(0, 2))]
        data = eol.join(lines) + (eol if rng.random() < 0.7 else b"")
        a, ea = run(data, p, "0")
        b, eb = run(data, p, "1")
        assert (a is None) == (b is None), (case, ea, eb)
        if a is not None:
            assert same(a, b), ("fasta", case)
            n_or += vs_oracle(b, data, p)
        if FILES:
            check_files(a, data, p, case)
        if SHARDED:
            check_sharded(a, data, p, case, rng)
        n_fa += 1
    else:
        eol = [b"\n", b"\r\n"][int(rng.integers(0, 2))]
        recs = []
        n_rec = int(rng.choice([1, 3, 40, 700]))
        damage = rng.random() < 0.6
        for i in range(n_rec):
            L = int(rng.choice([0, 1, 20, 150, 151, 3000])) if rng.random() < 0.3 else 150
            seq = bytes(rng.choice(alpha[:24], size=L))
            q = bytes(rng.choice(qual, size=L))
            if damage and rng.random() < 0.03:
                q = q[:-1] if L else b"I"                      # lengths differ
            if damage and rng.random() < 0.05 and L:
                q = bytes([64 if rng.random() < 0.5 else 43]) + q[1:]   # quality line starting with '@' / '+'
            rec = [b"@r%d" % i, seq, b"+" if rng.random() < 0.8 else b"+r%d" % i, q]
            if damage and rng.random() < 0.02:
                rec = rec[:int(rng.integers(1, 4))]            # truncated record
            recs.append(eol.join(rec))
            if damage and rng.random() < 0.03:
                recs.append(b"")                               # blank line between records
        data = eol.join(recs) + (eol if rng.random() < 0.7 else b"")
        a, ea = run(data, p, "0")
        b, eb = run(data, p, None)
        assert (a is None) == (b is None), (case, ea, eb)
        if a is not None:
            assert same(a, b), ("fastq", case)
            n_or += vs_oracle(b, data, p)
        else:
            n_fq_err += 1
        if FILES:
            check_files(a, data, p, case)
        if SHARDED:
            check_sharded(a, data, p, case, rng)
        n_fq += 1
print("fuzz_device_text: %d FASTA texts, %d FASTQ texts (%d refused by both parsers), %d also checked against the oracle, %d files, %d sharded runs: all agree"
      % (n_fa, n_fq, n_fq_err, n_or, n_files, n_sharded))
