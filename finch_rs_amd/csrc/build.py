#!/usr/bin/env python3
"""Build libfinch_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python finch_rs_amd/csrc/build.py [--force]

The hot kernel (fh_k2.hip) is compiled as FH_NPARTS translation units in parallel, one per share of
K = 1..32; everything is linked into finch_rs_amd/libfinch_hip.so (in-tree, travels with the repo).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libfinch_hip.so")
OBJ = os.path.join(HERE, "obj")
NPARTS = 4
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-cuda-compat"]
FLAGS += os.environ.get("FH_EXTRA_FLAGS", "").split()
# The sketch kernel's 32-position unrolled loop sits at the 128-VGPR limit of four waves per SIMD; LLVM's ILP-first
# scheduling strategy fits it without spills where the default one spills 11-21 registers (k = 21: +1 %, k = 24-31:
# +2.5 %, A/B on MI355X).
# The position loop is unrolled in full (`#pragma unroll`: every window offset a compile-time constant); LLVM honours the pragma
# only below -pragma-unroll-threshold (16 K instructions of IR before clean-up), which K = 22 crossed with round 5's admit path
# -- the loop came out rolled, its strings in scratch.
K2_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-pragma-unroll-threshold=200000"]
if "FH_K2_FLAGS" in os.environ:  # A/B builds of the sketch kernel only
    K2_FLAGS = os.environ["FH_K2_FLAGS"].split()
# The segment kernels of K = 25..32 (the last part; LDS-pipe bound: configs[2]'s k = 31) under LLVM's iterative ILP scheduler:
# the same instructions in another order, k = 29..31 +0.6-1.8 %, configs[2]'s launches +1.9 % (profiles/r05U_ab_iterative_ilp.txt);
# K <= 24 and the tile kernels gain nothing from it (k = 12, 20 lose half a per cent) and keep max-ilp.
K2S_FLAGS_LAST = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-mllvm", "-pragma-unroll-threshold=200000"]
if "FH_K2S_FLAGS_LAST" in os.environ:
    K2S_FLAGS_LAST = os.environ["FH_K2S_FLAGS_LAST"].split()
elif "FH_K2_FLAGS" in os.environ:
    K2S_FLAGS_LAST = K2_FLAGS
if "FH_OUT" in os.environ:  # an A/B build: its objects must not replace those libfinch_hip.so was linked from (tools/k2_regs.py --objects)
    OUT = os.environ["FH_OUT"]
    OBJ = os.path.join(HERE, "obj", "ab_" + os.path.splitext(os.path.basename(OUT))[0])

# the two-word segment kernels (K = 33..64) under max-ilp too: k = 33..64 +0.8-2.6 % over the default scheduler, no spills
# (profiles/r05W_ab_k2ws_sched.txt; iterative-ilp: +0-2.3 %)
K2WS_FLAGS = os.environ["FH_K2WS_FLAGS"].split() if "FH_K2WS_FLAGS" in os.environ else ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
# ... and the two-word tile kernels: k = 33 / 48 / 64 +3.2 / 2.9 / 3.7 % (FH_NO_SEG=1, profiles/r05W_ab_k2ws_sched.txt), no spills
K2W_FLAGS = os.environ["FH_K2W_FLAGS"].split() if "FH_K2W_FLAGS" in os.environ else ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
SOURCES = ["fh_core.h", "fh_device.h", "fh_kernels.h", "fh_k2_common.h", "fh_k2_lds.h", "fh_internal.h", "fh_options.h", "fh_options.cpp", "fh_k2b.hip", "fh_batch.hip", "fh_k2w.hip", "fh_k2.hip", "fh_k2s.hip", "fh_k2ws.hip", "fh_kernels.hip", "fh_big.hip", "fh_text.hip", "fh_bgzf.hip", "fh_api.hip", "fh_host.cpp",
           "fh_host_model.h", "fh_inflate.h", "fh_pargz.h", "fh_serial.cpp", os.path.join("..", "..", "include", "finch_host.h"),
           os.path.join("..", "..", "include", "finch_hip.h")]


def _newest_src():
    return max(os.path.getmtime(os.path.join(HERE, s)) for s in SOURCES)


def _run(cmd):
    r = subprocess.run(cmd, cwd=HERE, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _headers_mtime():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    inc = os.path.join(HERE, "..", "..", "include")
    hs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


_FORCE = False


def _compile(cmd):
    """One object: rebuilt only if it is older than its source or any header, or was made by another command line."""
    obj, src = cmd[-1], os.path.join(HERE, cmd[cmd.index("-c") + 1])
    stamp = obj + ".cmd"
    line = " ".join(cmd)
    if not _FORCE and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == line and \
            os.path.getmtime(obj) >= max(os.path.getmtime(src), _headers_mtime()):
        return ""
    out = _run(cmd)
    with open(stamp, "w") as f:
        f.write(line)
    return out


def build(force=False, verbose=False):
    global _FORCE
    _FORCE = force
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest_src():
        return OUT
    # several processes may get here at once (the ranks of a torch.distributed.run launch on a fresh checkout): one
    # builds, the others wait for the lock and find the library there
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest_src():
            return OUT
        return _build_locked(verbose)


def _build_locked(verbose):
    jobs = []
    for part in range(NPARTS):
        jobs.append([HIPCC] + FLAGS + K2_FLAGS + ["-DFH_PART=%d" % part, "-c", "fh_k2.hip", "-o", os.path.join(OBJ, "fh_k2_%d.o" % part)])
    for part in range(NPARTS):  # the batch form (many files per launch): the same per-tile code, the same flags
        jobs.append([HIPCC] + FLAGS + K2_FLAGS + ["-DFH_PART=%d" % part, "-c", "fh_k2b.hip", "-o", os.path.join(OBJ, "fh_k2b_%d.o" % part)])
    for part in range(NPARTS):  # the segment form of the sketch kernel
        jobs.append([HIPCC] + FLAGS + (K2S_FLAGS_LAST if part == NPARTS - 1 else K2_FLAGS) +
                    ["-DFH_PART=%d" % part, "-c", "fh_k2s.hip", "-o", os.path.join(OBJ, "fh_k2s_%d.o" % part)])
    for part in range(NPARTS):  # ... and of the two-word kernel
        jobs.append([HIPCC] + FLAGS + K2WS_FLAGS + ["-DFH_PART=%d" % part, "-c", "fh_k2ws.hip", "-o", os.path.join(OBJ, "fh_k2ws_%d.o" % part)])
    for part in range(NPARTS):  # K = 33..64
        jobs.append([HIPCC] + FLAGS + K2W_FLAGS + ["-DFH_PART=%d" % part, "-c", "fh_k2w.hip", "-o", os.path.join(OBJ, "fh_k2w_%d.o" % part)])
    jobs.append([HIPCC] + FLAGS + ["-c", "fh_kernels.hip", "-o", os.path.join(OBJ, "fh_kernels.o")])
    jobs.append([HIPCC] + FLAGS + ["-c", "fh_big.hip", "-o", os.path.join(OBJ, "fh_big.o")])
    jobs.append([HIPCC] + FLAGS + ["-c", "fh_text.hip", "-o", os.path.join(OBJ, "fh_text.o")])
    jobs.append([HIPCC] + FLAGS + ["-c", "fh_bgzf.hip", "-o", os.path.join(OBJ, "fh_bgzf.o")])
    jobs.append([HIPCC] + FLAGS + ["-c", "fh_api.hip", "-o", os.path.join(OBJ, "fh_api.o")])
    jobs.append([HIPCC] + FLAGS + ["-c", "fh_batch.hip", "-o", os.path.join(OBJ, "fh_batch.o")])
    jobs.append(["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-c", "fh_host.cpp", "-o", os.path.join(OBJ, "fh_host.o")])
    jobs.append(["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-c", "fh_serial.cpp", "-o", os.path.join(OBJ, "fh_serial.o")])
    jobs.append(["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-c", "fh_options.cpp", "-o", os.path.join(OBJ, "fh_options.o")])
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        outs = list(ex.map(_compile, jobs))
    if verbose:
        for o in outs:
            if o.strip():
                print(o)
    objs = [j[-1] for j in jobs]
    tmp = OUT + ".tmp.%d" % os.getpid()  # never let a reader dlopen a half-written library
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-lz", "-lpthread", "-ldl"])
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
