"""Randomised sketch / filter parameters through every route the host layer has for one input:
    python tools/fuzz_params.py [cases [seed]]      (on an MI355X)
host parser (FINCH_DEVICE_PARSE=0), device-side splitting (default), the full-size sketcher instead of the small one
(FINCH_NO_SMALL_SKETCHER=1), three handles on one device (sketch_stream_sharded), a gzip and a BGZF file through
sketch_files -- all must give the same Sketch (hashes, k-mers, counts, totals, filter parameters) or the same refusal
(strict mode with too few k-mers); with filtering off the Sketch is also held against the oracle's sketcher (first
final_size entries of its kmers_to_sketch).  The routes share the filters and the post filter; what differs is who parses, how the
text is cut, which sketcher size runs and whether partial sketches are merged -- none of which may show."""
import gzip, os, struct, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finch_rs_amd import host as H, sketch_schemes as S
import finch_rs_amd as F  # noqa: E402
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000


def bgzf(data, block=65280):
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        ch = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = co.compress(ch) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(ch), len(ch)))
    return b"".join(out)


def result(fn):
    try:
        sk = fn().sketch(0)
        return ("ok", sk.arrays[0].tobytes(), sk.arrays[1].tobytes(), sk.seq_length, sk.num_valid_kmers, repr(sk.filter_params))
    except S.FinchError as e:
        m = str(e)
        return ("err", m[m.index("had too few"):] if "had too few" in m else m)  # (the message starts with the sketch's name)


n_err = n_oracle = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    k = int(rng.choice([4, 11, 16, 21, 24, 31, 32, 33, 47, 64]))
    seed = int(rng.choice([0, 0, 42, 2**63 + 5]))
    gl = int(rng.choice([3000, 50000, 400000]))
    g = S.synth_genome_host(gl, int(rng.integers(1, 1000)))
    fastq = rng.random() < 0.6
    if fastq:
        n_reads, rl = int(rng.choice([30, 2000, 15000])), int(rng.choice([60, 150, 250]))
        reads = S.synth_reads_host(g, 0, n_reads, rl, int(rng.integers(1, 99)), int(rng.choice([0, 5000, 30000])), 500).reshape(n_reads, rl + 1)[:, :rl]
        data = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(reads[i]), b"I" * rl) for i in range(n_reads))
    else:
        w = int(rng.choice([60, 70, 10000]))
        data = b">g desc\n" + b"\n".join(bytes(g[j:j + w]) for j in range(0, gl, w)) + b"\n>second\n" + bytes(g[:gl // 3]) + b"\n"
    if rng.random() < 0.3:
        p = S.SketchParams.scaled(int(rng.choice([50, 1000])), k, float(rng.choice([0.001, 0.01, 0.2])), seed)
    else:
        final = int(rng.choice([1, 10, 1000, 3000]))
        p = S.SketchParams.mash(final * int(rng.choice([1, 1, 7, 200])), final, bool(rng.random() < 0.7), k, seed)
    fo = [None, True, False][int(rng.integers(0, 3))]
    f = H.FilterParams(fo, (int(rng.integers(1, 4)) if rng.random() < 0.4 else None, int(rng.integers(20, 200)) if rng.random() < 0.3 else None),
                       float(rng.choice([0.0, 0.05, 0.2, 1.0])), float(rng.choice([0.0, 0.1, 0.4])))
    F.debug_set(no_small_sketcher=None)
    F.debug_set(device_parse="0")
    ref = result(lambda: H.sketch_stream(data, "x", p, f))
    F.debug_set(device_parse=None)
    routes = {"device split": lambda: H.sketch_stream(data, "x", p, f)}  # (a small FASTA text: packed on the host while staged)

    def forced_splitter():
        F.debug_set(small_fasta_host="0")
        try:
            return H.sketch_stream(data, "x", p, f)
        finally:
            F.debug_set(small_fasta_host=None)
    routes["device splitter, small-file packing off"] = forced_splitter
    routes["sharded x3"] = lambda: H.sketch_stream_sharded(data, "x", p, f, [0, 0, 0], int(rng.integers(20000, 200000)))
    for name, fn in routes.items():
        r = result(fn)
        assert r == ref, (case, name, r[0], ref[0], r[1:] if r[0] == "err" else "", ref[1:] if ref[0] == "err" else "")
    F.debug_set(no_small_sketcher="1")
    r = result(lambda: H.sketch_stream(data, "x", p, f))
    F.debug_set(no_small_sketcher=None)
    assert r == ref, (case, "full-size sketcher", r[0], ref[0])
    for ext, img in ((".gz", gzip.compress(data, 1)), (".bgzf.gz", bgzf(data))):
        path = "/dev/shm/fuzz_params_%d%s" % (os.getpid(), ext)
        open(path, "wb").write(img)
        try:
            r = result(lambda: H.sketch_files([path], p, f, n_threads=2))
        finally:
            os.remove(path)
        assert r == ref, (case, ext, r[0], ref[0], r[1:] if r[0] == "err" else "", ref[1:] if ref[0] == "err" else "")
    filtering = fastq if fo is None else fo
    if ref[0] == "ok" and not filtering:
        o = O.OracleSketcher(O.SCALED if p.kind == "scaled" else O.MASH, p.kmers_to_sketch, k, seed, p.scale if p.kind == "scaled" else 0.001)
        assert o.sketch_stream(data) > 0
        okc, okm = o.to_vec()
        if p.kind != "scaled":
            okc, okm = okc[:p.final_size], okm[:p.final_size]
        assert ref[1] == okc.tobytes() and ref[2] == okm.tobytes(), (case, "oracle", len(ref[1]), okc.nbytes)
        assert (ref[3], ref[4]) == o.total_bases_and_kmers(), (case, "oracle totals")
        n_oracle += 1
    n_err += ref[0] == "err"
print("fuzz_params: %d parameter sets x 7 routes agree (%d of them refusals), %d also equal to the oracle's sketch" % (n_cases, n_err, n_oracle))
