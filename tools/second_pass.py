"""How much of a pass is spent converging the threshold?  Sketch the 10 Gbase bench stream, then push the same stream
again into the already-converged sketcher (threshold tight from the first base, every admitted k-mer already in the
table).  usage (GPU box): python tools/second_pass.py <k> <kmers_to_sketch>"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finch_rs_amd as F, finch_rs_amd.sketch_schemes as S
k, n = int(sys.argv[1]), int(sys.argv[2])
READ_LEN, GENOME_LEN, SEED = 150, 5_000_000, 20250620
n_reads = int(np.ceil(10e9 / READ_LEN)); nbytes = n_reads * 151
dg = F.DeviceBuffer(GENOME_LEN); dr = F.DeviceBuffer(nbytes + 64)
S.synth_genome_device(dg, GENOME_LEN, SEED); S.synth_reads_device(dr, dg, GENOME_LEN, 0, n_reads, READ_LEN, SEED, 10000, 500)
sk = F.SketchParams.mash(n, n, True, k, 0).create_sketcher()
for rep in range(2):
    sk.reset(); sk.sync()
    t0 = time.perf_counter(); sk.push_device(dr.ptr, nbytes); sk.sync(); t1 = time.perf_counter()
    sk.push_device(dr.ptr, nbytes); sk.sync(); t2 = time.perf_counter()
    print("k=%d n=%d first pass %.2f ms, second pass over the same data (threshold already converged) %.2f ms" % (k, n, (t1-t0)*1e3, (t2-t1)*1e3))
