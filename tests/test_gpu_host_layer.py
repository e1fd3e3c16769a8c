"""GPU tests of the host layer: finch::sketch_files / sketch_stream through the C++ mirror, against the
reference's CLI golden vectors and the oracle's own sketch_stream.  Run with -m gpu."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import finch_rs_amd as F

from finch_rs_amd import host as H
from finch_rs_amd import sketch_schemes as S
from finch_rs_amd.sketch_schemes import FinchError, SketchParams
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def oracle_sketch(data: bytes, kind, size, k, seed=0, scale=0.001):
    o = O.OracleSketcher(kind, size, k, seed, scale)
    fmt = o.sketch_stream(data)
    return o, fmt


def same(sk: H.Sketch, o: O.OracleSketcher, n=None):
    okc, okm = o.to_vec()
    if n is not None:
        okc, okm = okc[:n], okm[:n]
    assert np.array_equal(sk.arrays[0], okc) and np.array_equal(sk.arrays[1], okm)
    assert (sk.seq_length, sk.num_valid_kmers) == o.total_bases_and_kmers()


def test_cli_golden_through_sketch_files(golden_dir):
    # cli/tests/test_cli.rs:20-149 ("finch sketch --n-hashes 10 tests/data/query.fa -O"): the CLI oversketches
    # x200 (cli.rs:187-192) and FASTA input leaves filtering off (lib.rs:70-76)
    path = os.path.join(golden_dir, "query.fa")
    vec = json.load(open(os.path.join(golden_dir, "reference_vectors.json")))["test_cli_rs_99_143"]
    res = H.sketch_files([path], SketchParams.mash(10 * 200, 10, False, 21, 0), H.FilterParams(None, (None, None), 1.0 * 21 / 100, 1.0))
    sk = res.sketch(0)
    assert [h.kmer.decode() for h in sk.hashes] == vec["kmers"]
    assert sk.name == path and sk.seq_length == 405 and sk.num_valid_kmers == 339
    assert sk.filter_params.filter_on is False
    d = json.loads(res.to_json())
    assert (d["kmer"], d["alphabet"], d["sketchSize"], d["hashSeed"]) == (21, "ACGT", 10, 0)
    assert d["sketches"][0]["kmers"] == vec["kmers"] and len(d["sketches"][0]["hashes"]) == 10
    res2 = H.sketch_files([path], SketchParams.scaled(10, 21, 0.001, 0), H.FilterParams(None))
    assert [h.kmer.decode() for h in res2.sketch(0).hashes][:10] == vec["kmers"]


def make_fastq(n_reads, seed=3, rl=150, gl=100000):
    g = S.synth_genome_host(gl, seed)
    reads = S.synth_reads_host(g, 0, n_reads, rl, seed, 10000, 500).reshape(n_reads, rl + 1)[:, :rl]
    return b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(reads[i]), b"I" * rl) for i in range(n_reads)), g


def test_fastq_defaults_to_filtering_and_matches_oracle(tmp_path):
    fq, _ = make_fastq(20000)
    params = SketchParams.mash(2000, 100, False, 21, 0)
    filt = H.FilterParams(None, (None, None), 0.21, 0.1)
    p = tmp_path / "reads.fastq"
    p.write_bytes(fq)
    pz = tmp_path / "reads.fastq.gz"
    pz.write_bytes(gzip.compress(fq, 1))
    res = H.sketch_files([str(p), str(pz)], params, filt, n_threads=2)
    o, fmt = oracle_sketch(fq, O.MASH, 2000, 21)
    assert fmt == 2
    okc, okm = o.to_vec()
    a, ak = O.filter_strands(okc, okm, 0.1)
    cutoff = O.guess_filter_threshold(a, 0.21)
    b, bk = O.filter_abundance(a, ak, cutoff, None)
    for i in range(2):
        sk = res.sketch(i)
        assert sk.filter_params.filter_on is True and sk.filter_params.abun_filter == (cutoff, None)
        assert np.array_equal(sk.arrays[0], b[:100]) and np.array_equal(sk.arrays[1], bk[:100])
        assert (sk.seq_length, sk.num_valid_kmers) == o.total_bases_and_kmers()
    assert [res.sketch(i).name for i in range(2)] == [str(p), str(pz)]


def test_errors(tmp_path):
    with pytest.raises(FinchError, match="No such file or directory"):
        H.sketch_files([str(tmp_path / "nope.fa")], SketchParams.default(), H.FilterParams(False))
    p = tmp_path / "tiny.fa"
    p.write_bytes(b">a\nACGTACGTACGTACGTACGTACGTA\n")
    with pytest.raises(FinchError, match=r"had too few kmers \(\d+\) to sketch"):
        H.sketch_files([str(p)], SketchParams.default(), H.FilterParams(False))
    assert len(H.sketch_files([str(p)], SketchParams.mash(1000, 1000, True, 21, 0), H.FilterParams(False)).sketch(0).hashes) > 0


def test_batch_of_fastas_one_sketch_per_file_in_order(tmp_path):
    rng = np.random.default_rng(4)
    paths, datas = [], []
    for i in range(12):
        L = int(rng.integers(20000, 200000))
        seq = bytes(S.synth_genome_host(L, 100 + i))
        data = b">g%d\n" % i + b"\n".join(seq[j:j + 70] for j in range(0, L, 70)) + b"\n"
        p = tmp_path / ("g%02d.fa" % i)
        p.write_bytes(data)
        paths.append(str(p))
        datas.append(data)
    res = H.sketch_files(paths, SketchParams.default(), H.FilterParams(None), n_threads=4)
    assert len(res) == 12
    for i in range(12):
        o, fmt = oracle_sketch(datas[i], O.MASH, 1000, 21)
        sk = res.sketch(i)
        assert sk.name == paths[i]
        same(sk, o)


def _run_child(code, env):
    """env: options of the library (lower-case names: they travel in FH_DEBUG) and plain environment variables (upper-case)"""
    e = F.debug_env(**{k: v for k, v in env.items() if k.islower()})
    e.update({k: v for k, v in env.items() if not k.islower()})
    r = subprocess.run([sys.executable, "-c", code], env=e, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return r.stdout


def test_long_records_span_blocks_and_staging_slices():
    """a record longer than the host block / the staging slice: k-mers must span the cuts
    (FH_PUSH_CONTINUE + the K-1 byte carry).  Tiny buffers are forced through env knobs in a child process."""
    code = r'''
import numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F
from oracle import oracle as O
seq = bytes(S.synth_genome_host(300000, 9))
data = b">chr1 long\n" + b"\n".join(seq[j:j+61] for j in range(0, len(seq), 61)) + b"\n>chr2\n" + seq[1000:9000] + b"\nNNNN\n" + seq[:500] + b"\n"
res = H.sketch_stream(data, "mem", S.SketchParams.mash(500, 500, False, 31, 0), H.FilterParams(False))
o = O.OracleSketcher(O.MASH, 500, 31, 0); o.sketch_stream(data)
okc, okm = o.to_vec(); sk = res.sketch(0)
assert np.array_equal(sk.arrays[0], okc) and np.array_equal(sk.arrays[1], okm)
assert (sk.seq_length, sk.num_valid_kmers) == o.total_bases_and_kmers()
print("child ok")
'''
    for env in [{"block_bytes": "1000"}, {"stage_bytes": "4096"}, {"block_bytes": "7777", "stage_bytes": "5000"}]:
        assert "child ok" in _run_child(code, env)


def test_wide_kmers_span_staging_slices_on_every_path():
    """k = 34..64 carry up to 63 bytes from one staging slice to the next: records longer than a 4 KiB slice through the
    host parser (staged pushes: the carry sits in front of the staged data), through the device-side FASTA splitter and
    through the FASTQ path with its fallback, against the oracle.  (The room in front of the staging buffer was once 32
    bytes: k >= 34 wrote before the allocation.)"""
    code = r'''
import os, numpy as np
import finch_rs_amd as F
from finch_rs_amd import host as H, sketch_schemes as S
from oracle import oracle as O
seq = bytes(S.synth_genome_host(60000, 21))
fa = b">a long one\n" + b"\n".join(seq[j:j+70] for j in range(0, 40000, 70)) + b"\n>b\n" + seq[40000:52000] + b"\n"
reads = [seq[i * 700:i * 700 + (3000 if i % 5 == 0 else 150)] for i in range(60)]
fq = b"".join(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(reads))
for k in (33, 34, 48, 63, 64):
    p = S.SketchParams.mash(200, 200, True, k, 0)
    for data in (fa, fq):
        o = O.OracleSketcher(O.MASH, 200, k, 0)
        assert o.sketch_stream(data) > 0
        okc, okm = o.to_vec()
        for mode in ("0", "1", None):
            if mode is None:
                F.debug_set(device_parse=None)
            else:
                F.debug_set(device_parse=mode)
            if mode == "1" and data is fq:
                continue  # (device only: a 3000-base record does not fit a 4 KiB slice -- an error by design)
            sk = H.sketch_stream(data, "x", p, H.FilterParams(False)).sketch(0)
            assert np.array_equal(sk.arrays[0], okc) and np.array_equal(sk.arrays[1], okm), (k, mode)
            assert (sk.seq_length, sk.num_valid_kmers) == o.total_bases_and_kmers(), (k, mode)
print("child ok")
'''
    assert "child ok" in _run_child(code, {"stage_bytes": "4096"})
    assert "child ok" in _run_child(code, {"stage_bytes": "5001"})


def test_fastq_cut_off_inside_a_record_is_the_reference_error_on_every_path():
    """needletail refuses a FASTQ file that ends inside a record ("truncated FASTQ record"); the device-side splitter
    hands such a chunk to the host parser instead of sketching what is there"""
    g = bytes(S.synth_genome_host(5000, 3))
    whole = b"".join(b"@r%d\n" % i + g[i * 100:i * 100 + 150] + b"\n+\n" + b"I" * 150 + b"\n" for i in range(20))
    p = S.SketchParams.mash(100, 100, True, 21, 0)
    good = H.sketch_stream(whole, "x", p, H.FilterParams(False)).sketch(0)
    for tail in (b"@last\n", b"@last", b"@last\nACGTACGTACGTACGTACGTACGT\n", b"@last\nACGTACGTACGTACGTACGTACGT", b"@last\nACGT\n+"):
        for mode in ("0", None, "1"):
            if mode is None:
                F.debug_set(device_parse=None)
            else:
                F.debug_set(device_parse=mode)
            try:
                with pytest.raises(FinchError):
                    H.sketch_stream(whole + tail, "x", p, H.FilterParams(False))
            finally:
                F.debug_set(device_parse=None)
    # the last record without its quality line's newline, and with an empty quality line for an empty sequence, are fine
    for tail in (b"@last\nACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIII", b"@last\n\n+\n"):
        for mode in ("0", None):
            if mode is None:
                F.debug_set(device_parse=None)
            else:
                F.debug_set(device_parse=mode)
            try:
                sk = H.sketch_stream(whole + tail, "x", p, H.FilterParams(False)).sketch(0)
            finally:
                F.debug_set(device_parse=None)
            assert sk.seq_length == good.seq_length + (24 if b"ACGT" in tail else 0)


def test_device_side_fastq_parsing_matches_host_parser(tmp_path):
    """with option device_parse=1 the sequence lines of plain FASTQ are found on the device
    (fh_text.hip); same sketch, seq_length and numValidKmers as the host parser, incl. CRLF, a last line
    without newline, quality lines starting with '@', and chunks much smaller than the file."""
    code = r'''
import os, sys, numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F
from oracle import oracle as O
def vs_oracle(b, data, p, tag):
    """the device-split sketch against the oracle's own parser + sketcher (returns False if the oracle's parser rejects the text)"""
    o = O.OracleSketcher(O.MASH, p.kmers_to_sketch, p.kmer_length, 0)
    if o.sketch_stream(data) <= 0:
        return False
    okc, okm = o.to_vec()
    assert np.array_equal(b.arrays[0], okc) and np.array_equal(b.arrays[1], okm), tag
    assert (b.seq_length, b.num_valid_kmers) == o.total_bases_and_kmers(), (tag, b.seq_length, b.num_valid_kmers, o.total_bases_and_kmers())
    return True
rng = np.random.default_rng(5)
g = S.synth_genome_host(200000, 3)
nr, rl = 30000, 150
reads = S.synth_reads_host(g, 0, nr, rl, 3, 10000, 500).reshape(nr, rl + 1)[:, :rl]
def fq(eol, last_newline=True, varlen=False):
    out = []
    for i in range(nr):
        L = int(rng.integers(1, rl + 1)) if varlen else rl
        q = bytes(rng.choice(np.frombuffer(b"@+IJ#5", np.uint8), size=L))
        out.append(b"@r%d some text" % i + eol + bytes(reads[i][:L]) + eol + b"+" + (b"r%d" % i if i % 3 == 0 else b"") + eol + q + eol)
    data = b"".join(out)
    return data if last_newline else data[:-len(eol)]
p = S.SketchParams.mash(500, 500, False, 21, 0)
f = H.FilterParams(False)
for name, data in [("lf", fq(b"\n")), ("crlf", fq(b"\r\n")), ("nolast", fq(b"\n", False)), ("varlen", fq(b"\n", True, True))]:
    path = os.path.join(sys.argv[1], name + ".fastq")
    open(path, "wb").write(data)
    F.debug_set(device_parse="0")
    a = H.sketch_files([path], p, f).sketch(0)
    F.debug_set(device_parse="1")
    b = H.sketch_files([path], p, f).sketch(0)
    assert np.array_equal(a.arrays[0], b.arrays[0]) and np.array_equal(a.arrays[1], b.arrays[1]), name
    assert (a.seq_length, a.num_valid_kmers) == (b.seq_length, b.num_valid_kmers), (name, a.seq_length, b.seq_length)
    assert vs_oracle(b, data, p, name), name  # LF, CRLF, no final newline, ragged reads + '@'/'+' quality lines: all oracle-checked
# not 4-line FASTQ -> loud error
bad = os.path.join(sys.argv[1], "bad.fastq")
open(bad, "wb").write(b"@r1\nACGT\n+\nIIII\n\n@r2\nACGT\n+\nIIII\n" * 1000)
try:
    H.sketch_files([bad], p, f)
    raise SystemExit("expected an error")
except S.FinchError as e:
    assert "FASTQ" in str(e)
# default mode (option device_parse not set): the device pass is tried first and the same file falls back to the host parser
blank = os.path.join(sys.argv[1], "blank.fastq")   # real reads, a blank line after every 7th record
recs = fq(b"\n").split(b"\n@r")
open(blank, "wb").write(b"\n@r".join(r + (b"\n" if i % 7 == 3 else b"") for i, r in enumerate(recs)))
F.debug_set(device_parse="0")
a = H.sketch_files([blank], p, f).sketch(0)
F.debug_set(device_parse=None)
b = H.sketch_files([blank], p, f).sketch(0)
assert np.array_equal(a.arrays[0], b.arrays[0]) and (a.seq_length, a.num_valid_kmers) == (b.seq_length, b.num_valid_kmers)
assert vs_oracle(b, open(blank, "rb").read(), p, "blank lines between records")
F.debug_set(device_parse="1")
try:
    H.sketch_files([blank], p, f)
    raise SystemExit("expected an error")
except S.FinchError as e:
    assert "FASTQ" in str(e)
F.debug_set(device_parse=None)
# what needletail checks per record, the device pass checks too: blanks / tabs / an interior CR inside a sequence line
# (normalize(false) drops them, k-mers span them) and unequal sequence / quality lengths send the file to the host
# parser -- same sketch as the oracle, or the reference's error
recs = fq(b"\n").split(b"\n@r")
def damage(i, r):
    if i % 11 == 5:
        lines = r.split(b"\n")
        s_, q_ = lines[1], lines[3]
        if len(s_) > 20:
            cut = 7 + i % 9
            ch = [b" ", b"\t", b"\r"][i % 3]
            lines[1] = s_[:cut] + ch + s_[cut:]
            lines[3] = q_[:cut] + b"#" + q_[cut:]
        return b"\n".join(lines)
    return r
ws = os.path.join(sys.argv[1], "ws.fastq")
data = b"\n@r".join(damage(i, r) for i, r in enumerate(recs))
open(ws, "wb").write(data)
b = H.sketch_files([ws], p, f).sketch(0)   # default mode: device pass refuses, host parser reads
assert vs_oracle(b, data, p, "blanks inside sequence lines")
F.debug_set(device_parse="1")
try:
    H.sketch_files([ws], p, f)
    raise SystemExit("expected an error")
except S.FinchError as e:
    assert "FASTQ" in str(e)
F.debug_set(device_parse=None)
mm = os.path.join(sys.argv[1], "mismatch.fastq")
lines = fq(b"\n").split(b"\n")
lines[4 * 777 + 3] = lines[4 * 777 + 3][:-1]          # one quality line a byte short
open(mm, "wb").write(b"\n".join(lines))
for mode in (None, "1", "0"):
    if mode is None:
        F.debug_set(device_parse=None)
    else:
        F.debug_set(device_parse=mode)
    try:
        H.sketch_files([mm], p, f)
        raise SystemExit("expected an error (mode %r)" % mode)
    except S.FinchError as e:
        assert ("lengths differ" in str(e)) if mode != "1" else ("FASTQ" in str(e)), (mode, str(e))
F.debug_set(device_parse=None)
# ... and a well-formed file takes the device path by default with the host parser's result
good = os.path.join(sys.argv[1], "lf.fastq")
b = H.sketch_files([good], p, f).sketch(0)
F.debug_set(device_parse="0")
a = H.sketch_files([good], p, f).sketch(0)
assert np.array_equal(a.arrays[0], b.arrays[0]) and (a.seq_length, a.num_valid_kmers) == (b.seq_length, b.num_valid_kmers)
print("child ok")
'''
    for env in [{}, {"stage_bytes": "65536"}]:
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], env=e,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0 and "child ok" in r.stdout, r.stdout


def test_device_side_fasta_parsing_matches_host_parser(tmp_path):
    """with option device_parse=1 the sequence bytes of multi-line FASTA are found on the device
    (fh_text.hip, latest-event max-scan); same sketch, seq_length and numValidKmers as the host parser: LF / CRLF,
    ragged line lengths, '>' inside lines, blank lines, spaces and tabs, empty records, a last line without newline,
    many short records, 0xFF bytes, header and sequence lines longer than the staging chunk, chunks of 4 KiB"""
    code = r'''
import os, sys, numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F
from oracle import oracle as O
def vs_oracle(b, data, p, tag):
    """the device-split sketch against the oracle's own parser + sketcher (returns False if the oracle's parser rejects the text)"""
    o = O.OracleSketcher(O.MASH, p.kmers_to_sketch, p.kmer_length, 0)
    if o.sketch_stream(data) <= 0:
        return False
    okc, okm = o.to_vec()
    assert np.array_equal(b.arrays[0], okc) and np.array_equal(b.arrays[1], okm), tag
    assert (b.seq_length, b.num_valid_kmers) == o.total_bases_and_kmers(), (tag, b.seq_length, b.num_valid_kmers, o.total_bases_and_kmers())
    return True
rng = np.random.default_rng(11)
g = bytes(S.synth_genome_host(600000, 13))
def wrap(seq, w, eol):
    return eol.join(seq[i:i + w] for i in range(0, len(seq), w)) + eol
def contigs(n, eol, w, last_newline=True):
    cuts = sorted(rng.integers(0, len(g), n - 1).tolist())
    parts = [g[a:b] for a, b in zip([0] + cuts, cuts + [len(g)])]
    out = b"".join(b">contig%d len=%d" % (i, len(s)) + eol + (wrap(s, w, eol) if s else b"") for i, s in enumerate(parts))
    return out if last_newline else out.rstrip(b"\r\n")
cases = {
    "lf70": contigs(7, b"\n", 70),
    "crlf60": contigs(5, b"\r\n", 60),
    "nolast": contigs(3, b"\n", 80, False),
    "oneline": b">x\n" + g[:300000] + b"\n>y\n" + g[300000:] + b"\n",           # lines far longer than a chunk
    "longheader": b">" + b"h" * 20000 + b"\n" + wrap(g[:50000], 61, b"\n"),
    "many": b"".join(b">r%d\n" % i + g[i * 97:i * 97 + int(rng.integers(1, 200))] + b"\n" for i in range(4000)),
    "odd": (b">a desc > with gt\nACGT>ACGTAC\n\n  ACG TAC\tGT \r\nNNNN\n>\n>empty\n>b\n" + wrap(g[1000:9000], 33, b"\n") +
            b">c\n" + g[:500] + b"\xff" + g[500:1000] + b"\n>last\nACGTACGTACGTACGTACGTACGTACGT"),
}
p = S.SketchParams.mash(400, 400, True, 21, 0)
f = H.FilterParams(False)
for name, data in cases.items():
    path = os.path.join(sys.argv[1], name + ".fa")
    open(path, "wb").write(data)
    F.debug_set(device_parse="0")
    a = H.sketch_files([path], p, f).sketch(0)
    for mode in ("1", None):  # explicit, and the default (FASTA is split on the device unless told otherwise)
        if mode is None:
            F.debug_set(device_parse=None)
        else:
            F.debug_set(device_parse=mode)
        b = H.sketch_files([path], p, f).sketch(0)
        assert np.array_equal(a.arrays[0], b.arrays[0]) and np.array_equal(a.arrays[1], b.arrays[1]), name
        assert (a.seq_length, a.num_valid_kmers) == (b.seq_length, b.num_valid_kmers), (name, a.seq_length, b.seq_length, a.num_valid_kmers, b.num_valid_kmers)
        # every one of these texts is FASTA to the oracle's parser: CRLF, '>' inside lines, blank lines, spaces and tabs,
        # empty records, no final newline, 0xFF bytes, lines longer than a chunk -- device splitter vs the oracle directly
        assert vs_oracle(b, data, p, name), name
print("child ok")
'''
    for env in [{}, {"stage_bytes": "65536"}, {"stage_bytes": "4096"}]:
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], env=e,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0 and "child ok" in r.stdout, r.stdout


def test_device_side_fasta_parsing_random_text():
    """random FASTA-shaped text (arbitrary bytes, '>' anywhere, LF / CRLF, blank lines, ragged lines, empty records)
    through sketch_stream: device-side splitting == host parser, sketch and seq_length, for several k; repeated with
    4 KiB staging chunks so that the cuts land everywhere (the knob is read when the library loads: child process)"""
    code = r'''
import os, numpy as np
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F
from oracle import oracle as O
def vs_oracle(b, data, p, tag):
    """the device-split sketch against the oracle's own parser + sketcher (returns False if the oracle's parser rejects the text)"""
    o = O.OracleSketcher(O.MASH, p.kmers_to_sketch, p.kmer_length, 0)
    if o.sketch_stream(data) <= 0:
        return False
    okc, okm = o.to_vec()
    assert np.array_equal(b.arrays[0], okc) and np.array_equal(b.arrays[1], okm), tag
    assert (b.seq_length, b.num_valid_kmers) == o.total_bases_and_kmers(), (tag, b.seq_length, b.num_valid_kmers, o.total_bases_and_kmers())
    return True
n_oracle = 0
alpha = np.frombuffer(b"ACGTACGTACGTACGTacgtNnuU>>- \t\r\xff*", dtype=np.uint8)
for case in range(30):
    rng = np.random.default_rng(4200 + case)
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
    k = int(rng.choice([3, 16, 21, 31]))
    p = S.SketchParams.mash(50, 50, True, k, 0)
    f = H.FilterParams(False)
    F.debug_set(device_parse="0")
    a = H.sketch_stream(data, "x", p, f).sketch(0)
    F.debug_set(device_parse="1")
    b = H.sketch_stream(data, "x", p, f).sketch(0)
    assert np.array_equal(a.arrays[0], b.arrays[0]) and np.array_equal(a.arrays[1], b.arrays[1]), case
    assert (a.seq_length, a.num_valid_kmers) == (b.seq_length, b.num_valid_kmers), (case, a.seq_length, b.seq_length)
    n_oracle += vs_oracle(b, data, p, case)
    # the default: a text that fits the staging buffer is packed on the host while it is staged (fasta_small_on_host);
    # with the tiny staging buffers of the repeats it does not fit and goes to the device-side splitter
    F.debug_set(device_parse=None)
    c = H.sketch_stream(data, "x", p, f).sketch(0)
    assert np.array_equal(a.arrays[0], c.arrays[0]) and np.array_equal(a.arrays[1], c.arrays[1]), case
    assert (a.seq_length, a.num_valid_kmers) == (c.seq_length, c.num_valid_kmers), (case, a.seq_length, c.seq_length)
assert n_oracle == 30, n_oracle  # the text always begins with '>': the oracle's parser takes every case
print("child ok")
'''
    for env in [{}, {"stage_bytes": "4096"}, {"stage_bytes": "5001"}]:
        assert "child ok" in _run_child(code, env)


def test_bgzf_and_gzip_files_give_the_sketch_of_the_plain_file(tmp_path):
    """sketch_files on the same FASTQ as plain text, gzip and BGZF (members inflated by the call's read threads)"""
    import struct
    import zlib
    g = S.synth_genome_host(400_000, 5)
    reads = S.synth_reads_host(g, 0, 40000, 150, 5, 10000, 500).reshape(40000, 151)[:, :150]
    fq = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for i in range(len(reads)))

    def bgzf(data, block=65280):
        out = []
        for i in list(range(0, len(data), block)) + [None]:
            ch = b"" if i is None else data[i:i + block]
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            c = co.compress(ch) + co.flush()
            out.append(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(c) + 25) + c
                       + struct.pack("<II", zlib.crc32(ch), len(ch)))
        return b"".join(out)

    files = {"a.fastq": fq, "a.fastq.gz": gzip.compress(fq, 1), "a.fastq.bgz": bgzf(fq)}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    p = SketchParams.mash(1000, 1000, False, 21, 0)
    res = H.sketch_files([str(tmp_path / n) for n in files], p, H.FilterParams(None))
    a = [res.sketch(i) for i in range(3)]
    for s in a[1:]:
        assert np.array_equal(s.arrays[0], a[0].arrays[0]) and np.array_equal(s.arrays[1], a[0].arrays[1])
        assert s.seq_length == a[0].seq_length and s.num_valid_kmers == a[0].num_valid_kmers
    # a single BGZF file: the call's read threads all go to its members
    one = H.sketch_files([str(tmp_path / "a.fastq.bgz")], p, H.FilterParams(None)).sketch(0)
    assert np.array_equal(one.arrays[0], a[0].arrays[0])
    bad = bytearray(files["a.fastq.bgz"])
    bad[len(bad) // 3] ^= 0x10
    (tmp_path / "bad.bgz").write_bytes(bytes(bad))
    with pytest.raises(FinchError):
        H.sketch_files([str(tmp_path / "bad.bgz")], p, H.FilterParams(None))


def test_compressed_text_goes_through_the_device_side_splitters(tmp_path, monkeypatch):
    """gz input is inflated on the host and its text split on the device like plain text: same sketches as the host
    parser for FASTA.gz, strict FASTQ.gz, and a FASTQ.gz with blank lines (device pass rejects it -> the gzip stream is
    rewound and read by the host parser)"""
    g = S.synth_genome_host(300_000, 8)
    fa = b">a desc\n" + b"\n".join(g.tobytes()[i:i + 61] for i in range(0, 200_000, 61)) + b"\n>b\n" + g.tobytes()[200_000:] + b"\n"
    reads = S.synth_reads_host(g, 0, 20000, 150, 8, 10000, 500).reshape(20000, 151)[:, :150]
    fq = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for i in range(len(reads)))
    loose = fq.replace(b"\n@r777\n", b"\n\n@r777\n", 1)  # a blank line between two records: needletail accepts it
    assert loose != fq
    files = {"a.fa.gz": gzip.compress(fa, 1), "s.fq.gz": gzip.compress(fq, 1), "l.fq.gz": gzip.compress(loose, 1)}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    paths = [str(tmp_path / n) for n in files]
    p = SketchParams.mash(500, 500, False, 21, 0)
    F.debug_set(device_parse="0")
    want = H.sketch_files(paths, p, H.FilterParams(None))
    F.debug_set(device_parse=None)
    got = H.sketch_files(paths, p, H.FilterParams(None))
    for i in range(len(paths)):
        a, b = want.sketch(i), got.sketch(i)
        assert np.array_equal(a.arrays[0], b.arrays[0]) and np.array_equal(a.arrays[1], b.arrays[1]), paths[i]
        assert (a.seq_length, a.num_valid_kmers) == (b.seq_length, b.num_valid_kmers), paths[i]
    assert np.array_equal(got.sketch(1).arrays[0], got.sketch(2).arrays[0])  # the blank line changes nothing
    # device only: the loose file is an error, the strict ones are not
    F.debug_set(device_parse="1")
    H.sketch_files(paths[:2], p, H.FilterParams(None))
    with pytest.raises(FinchError):
        H.sketch_files(paths[2:], p, H.FilterParams(None))
    # a truncated gzip stream fails loudly on either route
    (tmp_path / "t.fq.gz").write_bytes(files["s.fq.gz"][:len(files["s.fq.gz"]) // 2])
    for mode in ("0", "1", None):
        if mode is None:
            F.debug_set(device_parse=None)
        else:
            F.debug_set(device_parse=mode)
        with pytest.raises(FinchError):
            H.sketch_files([str(tmp_path / "t.fq.gz")], p, H.FilterParams(None))


def test_oversketch_without_filtering_uses_the_small_sketcher_and_changes_nothing(tmp_path, monkeypatch):
    """Mash, filtering off: bottom final_size of bottom kmers_to_sketch == bottom final_size, counts included, so the
    host layer sketches final_size hashes directly -- the result (JSON too) must equal the oversketched one"""
    g = S.synth_genome_host(2_000_000, 21)
    fa = tmp_path / "g.fa"
    fa.write_bytes(b">g\n" + b"\n".join(g.tobytes()[i:i + 80] for i in range(0, len(g), 80)) + b"\n")
    reads = S.synth_reads_host(g, 0, 30000, 150, 21, 10000, 500).reshape(30000, 151)[:, :150]
    fq = tmp_path / "r.fq"
    fq.write_bytes(b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for i in range(len(reads))))
    p = SketchParams.mash(50_000, 700, True, 21, 0)  # no_strict: the abundance filter empties the genome's sketch
    for filt in (H.FilterParams(None), H.FilterParams(False), H.FilterParams(True, (2, None), 0.0, 0.0)):
        F.debug_set(no_small_sketcher="1")
        want = H.sketch_files([str(fa), str(fq)], p, filt)
        F.debug_set(no_small_sketcher=None)
        got = H.sketch_files([str(fa), str(fq)], p, filt)
        assert got.to_json() == want.to_json()
        for i in range(2):
            a, b = want.sketch(i), got.sketch(i)
            assert len(b.arrays[0]) <= 700
            assert np.array_equal(a.arrays[0], b.arrays[0]) and np.array_equal(a.arrays[1], b.arrays[1])
            assert (a.seq_length, a.num_valid_kmers, a.filter_params) == (b.seq_length, b.num_valid_kmers, b.filter_params)


def test_one_input_across_several_device_handles(tmp_path):
    """finch_sketch_file_sharded (north_star: one large input partitioned across the GPUs of a node, host merge): on the
    1-GPU box the handles all sit on device 0.  FASTQ (record-aligned chunks) and FASTA with records far longer than a
    chunk (line-aligned chunks + the k-1 base halo across every cut) must give the Sketch of the single-handle path
    and of the oracle: hashes, counts, extra_counts, k-mers, seq_length, numValidKmers, filters."""
    fq, g = make_fastq(40000, seed=8, gl=300000)
    pq = tmp_path / "reads.fastq"
    pq.write_bytes(fq)
    pz = tmp_path / "reads.fastq.gz"
    pz.write_bytes(gzip.compress(fq, 1))
    filt = H.FilterParams(None, (None, None), 0.21, 0.1)
    for k, n_eff, final in ((21, 20000, 200), (31, 500, 500)):
        params = SketchParams.mash(n_eff, final, True, k, 0)
        one = H.sketch_files([str(pq)], params, filt).sketch(0)
        o, fmt = oracle_sketch(fq, O.MASH, n_eff, k)
        okc, okm = o.to_vec()
        a, ak = O.filter_strands(okc, okm, 0.1)
        cutoff = O.guess_filter_threshold(a, 0.21)
        b, bk = O.filter_abundance(a, ak, cutoff, None)
        for path, devs, chunk in ((pq, [0, 0, 0], 1 << 20), (pq, [0, 0], 70000), (pz, [0, 0, 0, 0, 0], 300000), (pq, [0], 0)):
            sk = H.sketch_file_sharded(str(path), params, filt, devs, chunk).sketch(0)
            assert np.array_equal(sk.arrays[0], one.arrays[0]) and np.array_equal(sk.arrays[1], one.arrays[1]), (k, devs, chunk)
            assert (sk.seq_length, sk.num_valid_kmers) == (one.seq_length, one.num_valid_kmers) == o.total_bases_and_kmers()
            assert sk.filter_params.abun_filter == (cutoff, None) and sk.filter_params.filter_on is True
            assert np.array_equal(sk.arrays[0], b[:final]) and np.array_equal(sk.arrays[1], bk[:final])
    # FASTA: two chromosomes of 0.9 and 0.3 Mb (70-column lines, CRLF in the second) + short contigs, an N run, blank lines
    g = bytes(S.synth_genome_host(1_200_000, 21))
    fa = (b">chr1 long\n" + b"\n".join(g[j:j + 70] for j in range(0, 900000, 70)) + b"\n>chr2\r\n" +
          b"\r\n".join(g[j:j + 61] for j in range(900000, 1200000, 61)) + b"\r\n>c3\nACGTNNNNNNN" + g[5000:5100] + b"\n\n" +
          g[100:180] + b"\n>c4\n" + g[777:800] + b"\n>empty\n>c5\n" + g[:60])
    pa = tmp_path / "genome.fa"
    pa.write_bytes(fa)
    for k in (21, 32, 11, 64, 45):  # (the halo a cut hands over is k-1 bases: up to 63)
        params = SketchParams.mash(1000, 1000, False, k, 0)
        o, fmt = oracle_sketch(fa, O.MASH, 1000, k)
        assert fmt == 1
        one = H.sketch_files([str(pa)], params, H.FilterParams(None)).sketch(0)
        same(one, o)
        for devs, chunk in (([0, 0, 0], 4096), ([0, 0], 5001), ([0, 0, 0, 0], 1 << 16), ([0, 0, 0], 0)):
            sk = H.sketch_file_sharded(str(pa), params, H.FilterParams(None), devs, chunk).sketch(0)
            same(sk, o)
            assert sk.filter_params.filter_on is False and sk.name == str(pa)
    # in-memory image, scaled sketch
    res = H.sketch_stream_sharded(fa, "mem", SketchParams.scaled(100, 21, 0.01, 0), H.FilterParams(None), [0, 0, 0], 8192)
    o, _ = oracle_sketch(fa, O.SCALED, 100, 21, 0, 0.01)
    same(res.sketch(0), o)
    # FASTQ the device-side splitter does not take but needletail does (blank lines between records): the file goes
    # through one handle and the host parser, as it does in finch_sketch_files; what no parser takes is an error
    odd = tmp_path / "odd.fastq"
    odd.write_bytes(b"".join(b"@r%d\n" % i + g[i * 90:i * 90 + 120] + b"\n+\n" + b"I" * 120 + b"\n\n" for i in range(300)))
    pp = SketchParams.mash(100, 100, True, 21, 0)
    ref = H.sketch_files([str(odd)], pp, H.FilterParams(False)).sketch(0)
    for devs, chunk in (([0, 0], 4096), ([0, 0, 0], 0)):
        sk = H.sketch_file_sharded(str(odd), pp, H.FilterParams(False), devs, chunk).sketch(0)
        assert np.array_equal(sk.arrays[0], ref.arrays[0]) and np.array_equal(sk.arrays[1], ref.arrays[1])
        assert (sk.seq_length, sk.num_valid_kmers) == (ref.seq_length, ref.num_valid_kmers) and ref.seq_length == 300 * 120
    sk = H.sketch_stream_sharded(odd.read_bytes(), "mem", pp, H.FilterParams(False), [0, 0], 4096).sketch(0)
    assert np.array_equal(sk.arrays[0], ref.arrays[0])
    bad = tmp_path / "bad.fastq"
    bad.write_bytes(b"@r1\nACGT\n+\nIII\n@r2\nACGT\n+\nIIII\n" * 100)
    with pytest.raises(FinchError, match="FASTQ"):
        H.sketch_file_sharded(str(bad), SketchParams.mash(10, 10, True, 21, 0), H.FilterParams(False), [0, 0], 4096)
    with pytest.raises(FinchError, match="No such file"):
        H.sketch_file_sharded(str(tmp_path / "nope.fq"), SketchParams.default(), H.FilterParams(False), [0, 0], 0)


def test_fastq_text_in_memory_is_stripped_on_the_host_and_matches_the_device_splitter():
    """finch_sketch_buffer on FASTQ text with read threads to spare: headers, '+' lines and quality strings are dropped on the
    host (fh_fqstrip.h) and only the packed stream crosses the link -- same Sketch as through the device-side splitter and as
    the oracle's sketch_stream, for LF / CRLF text, a missing last newline, many small chunks; text that is not plain 4-line
    FASTQ goes to the host parser as before"""
    code = r'''
import os, numpy as np
import finch_rs_amd as F
from finch_rs_amd import host as H, sketch_schemes as S
from oracle import oracle as O
g = S.synth_genome_host(200_000, 5)
reads = S.synth_reads_host(g, 0, 30_000, 150, 3, 10000, 500).reshape(30_000, 151)[:, :150]
def fq(eol=b"\n", last=True, blank_at=None):
    recs = [b"@read%d extra" % i + eol + bytes(reads[i]) + eol + b"+" + eol + b"I" * 150 for i in range(len(reads))]
    if blank_at is not None:
        recs[blank_at] = recs[blank_at] + eol  # a blank line behind a record: needletail reads on, the strip must hand over
    return eol.join(recs) + (eol if last else b"")
p = S.SketchParams.mash(1000, 1000, False, 21, 0)
for eol, last in ((b"\n", True), (b"\r\n", True), (b"\n", False)):
    data = fq(eol, last)
    o = O.OracleSketcher(O.MASH, 1000, 21, 0); o.sketch_stream(data); okc, okm = o.to_vec()
    F.debug_set(fastq_host_strip="0")
    a = H.sketch_stream(data, "mem", p, H.FilterParams(False)).sketch(0)
    n0 = H.debug_fastq_host_strip()
    F.debug_set(fastq_host_strip="1")
    b = H.sketch_stream(data, "mem", p, H.FilterParams(False)).sketch(0)
    assert H.debug_fastq_host_strip() == n0 + 1
    for sk in (a, b):
        assert np.array_equal(sk.arrays[0], okc) and np.array_equal(sk.arrays[1], okm)
        assert (sk.seq_length, sk.num_valid_kmers) == o.total_bases_and_kmers()
# not plain 4-line FASTQ: the strip refuses, the parser that is the judge of it reads the text
data = fq(blank_at=777)
o = O.OracleSketcher(O.MASH, 1000, 21, 0); o.sketch_stream(data); okc, okm = o.to_vec()
n0 = H.debug_fastq_host_strip()
c = H.sketch_stream(data, "mem", p, H.FilterParams(False)).sketch(0)
assert H.debug_fastq_host_strip() == n0
assert np.array_equal(c.arrays[0], okc) and np.array_equal(c.arrays[1], okm) and (c.seq_length, c.num_valid_kmers) == o.total_bases_and_kmers()
print("child ok")
'''
    for env in ({}, {"fastq_strip_chunk": "70000"}, {"fastq_strip_chunk": "1000000", "read_threads": "3"}):
        assert "child ok" in _run_child(code, env)
