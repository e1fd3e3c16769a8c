import sys, time, numpy as np, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finch_rs_amd as F
import finch_rs_amd.sketch_schemes as S
from finch_rs_amd.sketch_schemes import KC_DTYPE, check
n_reads = 10_000_000
nbytes = n_reads * 151
dg = F.DeviceBuffer(5_000_000); dr = F.DeviceBuffer(nbytes + 64)
S.synth_genome_device(dg, 5_000_000, 1); S.synth_reads_device(dr, dg, 5_000_000, 0, n_reads, 150, 1, 10000, 500)
sk = F.SketchParams.mash(2_000_000, 2_000_000, True, 31, 0).create_sketcher()
sk.push_device(dr.ptr, nbytes)
n, _ = sk.finish()
for rep in range(3):
    t = [time.perf_counter()]
    hs = np.empty(n, np.uint64); cs = np.empty(n, np.uint32); es = np.empty(n, np.uint32); km = np.empty((n, 31), np.uint8); ps = np.empty(n, np.uint64)
    t.append(time.perf_counter())
    check(sk._L.fh_copy_out(sk._h, hs.ctypes.data_as(C.c_void_p), cs.ctypes.data_as(C.c_void_p), es.ctypes.data_as(C.c_void_p), km.ctypes.data_as(C.c_void_p), ps.ctypes.data_as(C.c_void_p)))
    t.append(time.perf_counter())
    check(sk._L.fh_copy_out(sk._h, hs.ctypes.data_as(C.c_void_p), cs.ctypes.data_as(C.c_void_p), es.ctypes.data_as(C.c_void_p), km.ctypes.data_as(C.c_void_p), ps.ctypes.data_as(C.c_void_p)))
    t.append(time.perf_counter())
    kc = np.empty(n, dtype=KC_DTYPE); kc["hash"], kc["count"], kc["extra_count"] = hs, cs, es
    t.append(time.perf_counter())
    print("n=%d alloc %.2f  copy_out(first touch) %.2f  copy_out(again) %.2f  structured %.2f ms" % ((n,) + tuple((t[i+1]-t[i])*1e3 for i in range(4))))
