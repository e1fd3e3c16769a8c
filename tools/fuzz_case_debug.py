"""One case of tests/test_gpu_fuzz.py replayed with diagnostics: which mode / rep differs from the oracle, how (hashes missing,
counts off), and the sketcher's counters.   python tools/fuzz_case_debug.py <seed> <case>   (GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import finch_rs_amd as F
from finch_rs_amd import _lib
from oracle import oracle as O
import test_gpu_fuzz as T

seed0, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0 + case)
k = int(rng.choice([1, 2, 3, 5, 8, 11, 15, 16, 17, 21, 21, 21, 24, 27, 31, 31, 32, 33, 40, 48, 55, 63, 64]))
kind = "mash" if rng.random() < 0.6 else "scaled"
size = int(rng.choice([0, 1, 7, 100, 1000, 1000, 2999, 3001, 12000]))
scale = float(rng.choice([1.0, 0.5, 0.01, 0.001]))
seed = int(rng.choice([0, 0, 42, 2**63 + 12345]))
inflight = int(rng.choice([0, 0, 4096, 16384, 1 << 20]))
genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.choice([3000, 60000, 400000])))
n_rec = int(rng.choice([0, 1, 30, 400, 3000]))
maxlen = int(rng.choice([40, 200, 5000]))
if n_rec * maxlen > 3_000_000:
    n_rec = 3_000_000 // maxlen
if kind == "scaled" and scale >= 0.5 and n_rec * maxlen > 600_000:
    n_rec = 600_000 // maxlen
recs = [T.rand_record(rng, genome, maxlen) for _ in range(n_rec)]
params = (F.SketchParams.mash(size, size, True, k, seed) if kind == "mash" else F.SketchParams.scaled(size, k, scale, seed))
stage = int(rng.choice([0, 8192, 70000]))
sk = params.create_sketcher(max_launch=inflight, stage_bytes=stage)
form = [None, "1", "0"][case % 3]
if form is not None:
    F.debug_set(read_first=form)
print(dict(k=k, kind=kind, size=size, seed=seed, inflight=inflight, n_rec=n_rec, maxlen=maxlen, stage=stage, form=form, total_bytes=sum(map(len, recs))))
forced = os.environ.get("MODE")
for rep in range(2):
    ora = O.OracleSketcher(O.MASH if kind == "mash" else O.SCALED, size, k, seed, scale)
    for r in recs:
        ora.process(r)
    mode = str(rng.choice(["one_block", "per_record", "cut_records", "resident"]))
    if forced:
        mode = forced
    if mode == "one_block":
        sk.push_block(b"".join(r + b"\x00" for r in recs))
    elif mode == "per_record":
        for r in recs:
            sk.process(r)
    elif mode == "cut_records":
        import ctypes as C
        L = _lib.load()
        for r in recs:
            cuts = sorted(set(int(x) for x in rng.integers(0, len(r) + 1, size=int(rng.integers(0, 3)))))
            pieces = [r[a:b] for a, b in zip([0] + cuts, cuts + [len(r)])]
            for i, p in enumerate(pieces):
                last = i == len(pieces) - 1
                blk = np.frombuffer(p + (b"\x00" if last else b""), dtype=np.uint8)
                if len(blk) == 0:
                    continue
                _lib.check(L.fh_push_block_ex(sk._h, blk.ctypes.data_as(C.c_void_p), len(blk), 1 if i > 0 else 0))
    else:
        packed = b"".join(r.translate(bytes.maketrans(b"", b""), b" \t\r\n") + b"\x00" for r in recs)
        buf = F.DeviceBuffer(len(packed) + 64)
        buf.upload(np.frombuffer(packed, dtype=np.uint8))
        sk.push_device(buf.ptr, len(packed))
        sk.sync()
    kc, km, pos = sk.to_arrays()
    okc, okm = ora.to_vec()
    dh, oh = set(kc["hash"].tolist()), set(okc["hash"].tolist())
    common = dh & oh
    dd = {int(h): (int(c), int(e)) for h, c, e in zip(kc["hash"], kc["count"], kc["extra_count"])}
    od = {int(h): (int(c), int(e)) for h, c, e in zip(okc["hash"], okc["count"], okc["extra_count"])}
    off = sum(1 for h in common if dd[h] != od[h])
    print("rep %d mode %-11s: device %d oracle %d hashes, missing on device %d, extra %d, common with other counts %d, total_kmers %d vs %d, dbg %s"
          % (rep, mode, len(kc), len(okc), len(oh - dh), len(dh - oh), off, sk.finish()[1], ora.total_bases_and_kmers()[1], sk.debug_counters()))
    if oh - dh:
        miss = sorted(oh - dh)
        print("   smallest missing %d (rank %d of oracle), device max %d, oracle max %d" % (miss[0], sorted(oh).index(miss[0]), max(dh) if dh else -1, max(oh)))
    sk.reset()
