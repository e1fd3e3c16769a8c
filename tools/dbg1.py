import numpy as np, sys
sys.path.insert(0,'.')
import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S
from oracle import oracle as O
gl, nr, rl, seed = 200000, 8000, 150, 5
g = S.synth_genome_host(gl, seed)
reads = S.synth_reads_host(g, 0, nr, rl, seed, 10000, 500)
size, scale = 100, 0.25
for ml in [8192, 1<<20]:
    sk = F.SketchParams.scaled(size, 21, scale, 0).create_sketcher(max_launch=ml)
    sk.push_block(reads)
    kc, km, pos = sk.to_arrays()
    ora = O.OracleSketcher(O.SCALED, size, 21, 0, scale); ora.process_packed(reads, 0)
    okc, okm = ora.to_vec()
    h = kc["hash"]; oh = okc["hash"]
    print("ml", ml, "ours", len(h), "oracle", len(oh), "unique ours", len(np.unique(h)), sk.debug_counters())
    extra = np.setdiff1d(h, oh); missing = np.setdiff1d(oh, h)
    print(" extra", len(extra), "missing", len(missing))
    if len(extra):
        idx = np.isin(h, extra)
        print(" extra counts", kc["count"][idx][:10], "pos", pos[idx][:10], "kmers", [bytes(k).decode() for k in km[idx][:3]])
        print(" max_hash", (2**64-1)//4, "extra min/max", extra.min(), extra.max())
    vals, cnt = np.unique(h, return_counts=True)
    print(" dups", (cnt>1).sum())
    common = np.intersect1d(h, oh)
    a = kc[np.isin(h, common)]; b = okc[np.isin(oh, common)]
    if len(np.unique(h)) == len(h):
        print(" count mismatches among common", int((a["count"] != b["count"]).sum()))
