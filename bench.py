#!/usr/bin/env python3
"""bench.py -- bases/sec sketched (k=21, n=1000) on N x MI355X, with the kernel's HBM roofline fraction, the CPU baseline
timed beside it and a self-check of the sketch against the oracle's golden fingerprint (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2|c5] [--gbases G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

How the N GPUs are driven.  Launched by torch.distributed.run (WORLD_SIZE in the environment): one process per GPU, rank r
sketches read block r, rank 0 gathers the <= n-record partial sketches and merges them on the host.  Launched plainly with
--gpus N > 1: ONE process and one fh_sketch_device_blocks call per step -- the library runs one host thread and one sketcher
handle per device (the reference's own shape: finch is one process, lib.rs:34-36) and merges the partial sketches.  Either way there is no
data-path collective: read blocks are independent and the merge is O(N n).

Workloads (synthetic 150 bp FASTQ-shaped read sets of SURVEY.md 8d M4, already resident in HBM as the packed sequence stream
-- 150 bases + 1 breaker byte per read -- when the timed region starts; Mash sketch k=21, kmers_to_sketch=1000, seed 0):
  c4  BASELINE.json configs[3], the default for EVERY N: 50 Gbase IN TOTAL, the reads split into N contiguous read blocks
      (shard_bounds), one per GPU, partial sketches merged on the host -- strong scaling; at N = 1 the whole 50.3 GB stream
      sits on the one GPU.  This is the workload north_star's 1/2/4/8-GPU target is stated on.
  c2  BASELINE.json configs[1]: 10 Gbase per GPU (weak scaling when N > 1).
  c5  BASELINE.json configs[4]: a batch of synthetic RefSeq-sized FASTA files (log-uniform 1-10 Mb, 70-column lines) in
      tmpfs through ONE finch_sketch_files call with devices = [0..N-1] (lib.rs:29-49); --files F (default 10000, cut to what
      the tmpfs holds and said so).  Under torch.distributed.run rank r takes files r, r+N, ...
A step = one full pass: reset, sketch every base of the block, finish (bottom-n select, copy-out of the <= 1000 records to
the host) and -- for N > 1 -- the host-side merge, which runs one step behind the sketching on a thread of its own (all K merges
complete inside the timed region).  value = total bases of all GPUs * K / max-over-ranks time.

Self-check: the seven-number fingerprint of the final sketch (+ total_kmers) is compared with
tests/golden/config_fingerprints.json, which the ORACLE produced on the CPU (tests/golden/make_config_fingerprints.py);
"sketch_check.matches_golden" is true / false / null (no golden for a non-default size), and the exit code is 3 when false.

After the timed region an N = 1 run also measures, OUTSIDE `value` (key "extras"): configs[1] (the first 10 Gbase of the same
stream), k=31, configs[2] (k=31, 2 M hashes, host filters), the CLI-default oversketch (n=200 000), a two-word k-mer length,
the end-to-end rate from FASTQ text in host memory, compressed input and a batch of FASTA files.  --no-extras skips them.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20250620
GENOME_LEN = 5_000_000
READ_LEN = 150
REC = READ_LEN + 1
SUB_PPM, N_PPM = 10_000, 500
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
C5_SAMPLE = 256        # files 0..255 of the batch carry a golden fingerprint


_SHARED = {}  # inherited by fork()ed workers: no pickling of the sample


def _usable_cpus() -> int:
    """hardware threads this process may actually use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def _oracle_shard_job(job):
    from oracle import oracle as O
    i, per_bytes, n, k = job
    data = _SHARED["big"][i * per_bytes:(i + 1) * per_bytes]
    o = O.OracleSketcher(O.MASH, n, k, 0)
    o.process_packed(data, 0)
    return len(o.to_vec()[0])


def _write_fasta_job(job):
    """one synthetic genome as a 70-column FASTA file (log-uniform 1..10 Mb from the per-file seed)"""
    from finch_rs_amd import sketch_schemes as S
    d, i = job
    path = os.path.join(d, "g%05d.fa" % i)
    with open(path, "wb") as f:
        f.write(S.synth_fasta_file(i, SEED))
    return path, S.synth_fasta_length(i, SEED)


def _oracle_file_job(path):
    """one file through the oracle's sketch_stream (lib.rs:51-94), as one rayon task of sketch_files would (lib.rs:34-36)"""
    from oracle import oracle as O
    with open(path, "rb") as f:
        data = f.read()
    o = O.OracleSketcher(O.MASH, 1000, 21, 0)
    o.sketch_stream(data)
    return o.total_bases_and_kmers()[0]


def c5_cpu_baseline(paths, lens, max_files=256):
    """BASELINE configs[4]'s CPU side: the oracle's sketch_stream over the first files of the SAME batch, one
    file per task on every core this process is granted -- for a batch of files that IS the reference's behaviour
    (sketch_files is a rayon par_iter over the file names, lib.rs:34-36)"""
    import multiprocessing as mp
    from oracle import oracle as O
    flags = O.use_native()
    cores = max(1, min(_usable_cpus(), 256))
    sample = list(range(min(len(paths), max(max_files, 2 * cores) if len(paths) >= 2 * cores else len(paths))))
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_oracle_file_job, [paths[0]] * cores, chunksize=1)  # start the workers, page the files' directory in
        c0 = time.perf_counter()
        got = pool.map(_oracle_file_job, [paths[i] for i in sample], chunksize=1)
        ct = time.perf_counter() - c0
    bases = sum(lens[i] for i in sample)
    return {"value": round(bases / ct, 1), "unit": "bases/s", "cores": cores, "kind": "port", "files_per_s": round(len(sample) / ct, 1),
            "sample": "the first %d files of the same batch (%.0f Mbases) through the oracle's sketch_stream, one file per task on %d "
                      "worker processes = the reference's behaviour for a batch (rayon par_iter over files, lib.rs:34-36); gcc %s"
                      % (len(sample), bases / 1e6, cores, flags)}


def _h2d_peak_gbs(dev=0, mib=32, reps=16):
    """this box's pinned host-to-device copy rate (GB/s): copies of `mib` MiB on two streams at once (what a batch's workers do),
    best of four rounds -- what a batch of files is held against"""
    try:
        import torch
        hs = [torch.empty(mib << 20, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        ds = [torch.empty(mib << 20, dtype=torch.uint8, device="cuda:%d" % dev) for _ in range(2)]
        st = [torch.cuda.Stream(dev) for _ in range(2)]
        best = 0.0
        for it in range(4):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                for i in range(2):
                    with torch.cuda.stream(st[i]):
                        ds[i].copy_(hs[i], non_blocking=True)
            torch.cuda.synchronize(dev)
            best = max(best, 2 * reps * (mib << 20) / (time.perf_counter() - t0) / 1e9)
        return round(best, 2)
    except Exception:  # noqa: BLE001
        return None


def c5_roofline(kernel_ms, launches, positions, wall_ms, n_gpus, batch=None, live=None, h2d=None, text_bytes=None):
    """the batch's roofline block: the sketch kernel over ALL files (sum of its launches' HIP-event times on the workers'
    streams, finch_debug_kernel_times) against the HBM peak -- algorithmic bytes as SURVEY 8d counts them, 1 per k-mer start
    position --, how much of the call the GPU spent in it, and what the call's other two resources did: the bytes that cross the
    PCIe link (0.375 per position in the two-bit form the workers stage the files in, 1 with option batch_two_bit=0) against
    the link's measured rate, and the 16 host cores that read and pack the files"""
    import finch_rs_amd as F
    two_bit = F.get_option("batch_two_bit") != "0"
    per_pos = 0.375 if two_bit else 1.0
    ach = positions / 1e9 / (kernel_ms / 1e3) if kernel_ms > 0 else 0.0
    link = per_pos * positions / 1e9 / (wall_ms / 1e3) / max(n_gpus, 1) if wall_ms > 0 else 0.0
    out = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
           "traffic": None if not live else round(live["hbm_bytes_per_position"] * positions / max(launches, 1), 1),
           "kernel": "k2_batch<21> (fh_k2b.hip: the files a worker has staged, sketched by ONE launch; finished by one k_batch_epilogue "
                     "launch, a workgroup per file)", "launches": int(launches),
           "avg_launch_ms": round(kernel_ms / max(launches, 1), 4), "alg_bytes_per_launch": int(positions / max(launches, 1)),
           "input_bytes_per_position_in_hbm": per_pos,
           "kernel_ms_total": round(kernel_ms, 3), "call_ms_total": round(wall_ms, 3),
           "kernel_share_of_call": round(kernel_ms / max(wall_ms * n_gpus, 1e-9), 4),
           "pcie": {"h2d_peak_gbs": h2d, "achieved_gbs": round(link, 2), "frac": round(link / h2d, 3) if h2d else None,
                    "bytes": ("the files in the two-bit form (fh_batch_submit_packed): 2 bits of code + 1 'is a base' bit per k-mer start "
                              "position" if two_bit else "the files' packed streams, 1 byte per k-mer start position")
                             + ", in copies of up to 32 MiB per worker"},
           "binding_resource": ("the host: the workers (one per core the cgroup grants) read every file from the page cache and pack it into the "
                                "two-bit form at text_gbs_per_worker each; the link carries pcie.frac of what it could, the sketch kernels "
                                "take kernel_share_of_call of the call") if two_bit else
                               ("one PCIe link per GPU: the packed streams of the files cross it at pcie.achieved_gbs of the pcie.h2d_peak_gbs "
                                "this box's link copies (measured in this run); the sketch kernels take kernel_share_of_call of the call")}
    if text_bytes and wall_ms > 0:
        workers = max(1, min(_usable_cpus(), 16 * max(n_gpus, 1)))
        out["host"] = {"workers": workers, "text_gbs_total": round(text_bytes / 1e9 / (wall_ms / 1e3), 2),
                       "text_gbs_per_worker": round(text_bytes / 1e9 / (wall_ms / 1e3) / workers, 2),
                       "what": "FASTA text read from the page cache (pread, 256 KiB pieces) and packed (line ends out, two-bit form) per second of the call"}
    if batch is not None:
        out["files_taken_many_per_launch"], out["files_through_own_sketcher"] = batch
    if live:
        out["traffic_source"] = live["source"]
        out["pmc_live"] = live
    elif LIVE_PMC_WHY:
        out["traffic_source"] = "not collected: " + LIVE_PMC_WHY
    return out


def _pmc_derived(key):
    """what the committed rocprofv3 PMC passes say about the kernel's binding resource (profiles/pmc_summary.json, written
    by tools/pmc_summary.py from the counter files named there); None if no set was collected for this workload"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json"))).get(key)
    except Exception:
        return None


LIVE_PMC_WHY = None  # why the last _live_pmc returned None


def _live_pmc(child_args, timeout_s=150, kernels=("k2_sketch",)):
    """HBM bytes of the sketch kernel's launches from rocprofv3 PMC counters collected NOW, by this run: two child runs of this
    very command (--steps 1 --warmup 0, nothing else measured), one `--pmc` pass per counter group with --kernel-trace only
    (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass), FETCH_SIZE doubled per that guide's gfx950 correction for
    16 B/lane streaming reads, WRITE_SIZE as reported.  Returns None (the caller falls back on the committed passes and says so)
    if rocprofv3 is missing, fails, or does not finish in time -- or if this process is itself being profiled."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    global LIVE_PMC_WHY
    LIVE_PMC_WHY = None
    if shutil.which("rocprofv3") is None or any(k.startswith("ROCP") for k in os.environ):
        LIVE_PMC_WHY = "rocprofv3 not on PATH, or this process is itself being profiled"
        return None
    sums, disp, positions, t0 = {}, {}, None, time.perf_counter()
    base = tempfile.mkdtemp(prefix="fh_live_pmc_", dir="/tmp")
    try:
        for i, ctrs in enumerate((["GRBM_GUI_ACTIVE", "FETCH_SIZE"], ["WRITE_SIZE", "SQ_INSTS_VALU"],
                                  ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"])):
            d = os.path.join(base, "p%d" % i)
            cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline", "--no-live-pmc"] + child_args
            env = dict(os.environ, TMPDIR="/tmp")
            left = timeout_s - (time.perf_counter() - t0)
            if left < 20:
                if i == 2:
                    break
                LIVE_PMC_WHY = "out of its %d s" % timeout_s
                return None
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, start_new_session=True)
            try:
                out, _ = pr.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)  # exactly the process group started here
                pr.wait()
                if i == 2:
                    break  # (the LDS / wait counters are a bonus: the traffic figure stands without them)
                LIVE_PMC_WHY = "pass %d did not finish in time" % i
                return None
            if pr.returncode != 0:
                if i == 2:
                    break  # (the LDS / wait counters are a bonus: the traffic figure stands without them)
                LIVE_PMC_WHY = "pass %d: the profiled child run ended with code %d" % (i, pr.returncode)
                return None
            for line in out.splitlines():
                if line.startswith("{"):
                    r = json.loads(line)["roofline"]
                    positions = r["alg_bytes_per_launch"] * r["launches"]
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if any(kn in row["Kernel_Name"] for kn in kernels):
                        sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                        disp[row["Counter_Name"]] = disp.get(row["Counter_Name"], 0) + 1
        if not positions or "FETCH_SIZE" not in sums or "WRITE_SIZE" not in sums:
            LIVE_PMC_WHY = "no counters of %s in the profiler's output (positions %s, counters %s)" % ("/".join(kernels), positions, sorted(sums))
            return None
        hbm = 2.0 * sums["FETCH_SIZE"] * 1024.0 + sums["WRITE_SIZE"] * 1024.0
        return {"hbm_bytes_per_position": round(hbm / positions, 4), "fetch_size_kb": sums["FETCH_SIZE"], "write_size_kb": sums["WRITE_SIZE"],
                "valu_per_wave_iter": round(sums["SQ_INSTS_VALU"] / (positions / 64.0), 2) if "SQ_INSTS_VALU" in sums else None,
                "cycles_per_wave_iter": round(sums["GRBM_GUI_ACTIVE"] / 8 * 1024 / (positions / 64.0), 1) if "GRBM_GUI_ACTIVE" in sums else None,
                "cycles_per_valu_inst": round(sums["GRBM_GUI_ACTIVE"] / 8 * 1024 / sums["SQ_INSTS_VALU"], 3) if sums.get("SQ_INSTS_VALU") and "GRBM_GUI_ACTIVE" in sums else None,
                "lds_active_per_wave_iter": round(sums["SQ_LDS_IDX_ACTIVE"] / (positions / 64.0), 2) if "SQ_LDS_IDX_ACTIVE" in sums else None,
                "lds_bank_conflict_per_wave_iter": round(sums["SQ_LDS_BANK_CONFLICT"] / (positions / 64.0), 2) if "SQ_LDS_BANK_CONFLICT" in sums else None,
                "wait_any_frac_of_wave_cycles": round(sums["SQ_WAIT_ANY"] / sums["SQ_WAVE_CYCLES"], 3) if sums.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in sums else None,
                "positions": int(positions), "dispatches": disp.get("FETCH_SIZE"), "seconds": round(time.perf_counter() - t0, 1),
                "source": "live: rocprofv3 --pmc {GRBM_GUI_ACTIVE FETCH_SIZE | WRITE_SIZE SQ_INSTS_VALU | SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY} --kernel-trace over three child runs "
                          "of this command (--steps 1), summed over the %s* dispatches; HBM bytes = 2 x FETCH_SIZE (gfx950 " % "* / ".join(kernels) +
                          "correction for 16 B/lane streaming reads) + WRITE_SIZE"}
    except Exception as e:  # noqa: BLE001 -- a profiler hiccup must not take the bench line with it
        LIVE_PMC_WHY = "%s: %s" % (type(e).__name__, e)
        return None
    finally:
        shutil.rmtree(base, ignore_errors=True)


def _golden(key):
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "config_fingerprints.json"))).get(key)
    except Exception:
        return None


FP_KEYS = ("n_hashes", "min_hash", "max_hash", "hash_xor", "count_sum", "extra_sum", "kmer_byte_sum", "total_kmers")


def fingerprint(kc, km, total_kmers):
    return {"n_hashes": int(len(kc)), "min_hash": int(kc["hash"][0]) if len(kc) else None,
            "max_hash": int(kc["hash"][-1]) if len(kc) else None,
            "hash_xor": int(np.bitwise_xor.reduce(kc["hash"])) if len(kc) else 0,
            "count_sum": int(kc["count"].astype(np.uint64).sum()),
            "extra_sum": int(kc["extra_count"].astype(np.uint64).sum()),
            "kmer_byte_sum": int(np.asarray(km).astype(np.uint64).sum()),
            "total_kmers": int(total_kmers)}


def check_golden(fp, key):
    """-> (fingerprint + matches_golden, ok) ; matches_golden is None when the golden file has no entry for this workload"""
    g = _golden(key)
    out = dict(fp)
    out["golden"] = ("tests/golden/config_fingerprints.json[%s] (oracle, CPU)" % key) if g else None
    out["matches_golden"] = None if g is None else all(fp[k] == g[k] for k in FP_KEYS)
    return out, out["matches_golden"] is not False


class MergePipe:
    """Launched by torch.distributed.run: the gather + host-side merge of a step's partial sketches, one step behind the
    sketching -- a worker thread takes each step's partial sketch in order, ships it to rank 0 (one small tensor per rank) and
    rank 0's merges them while the GPUs are already on the next step.  flush() returns when everything handed over has been
    merged -- the timed region ends with one, so all K merges are inside it; at most two steps may be in flight."""

    def __init__(self, fn, init=None):
        import queue
        import threading
        self.fn, self.init, self.last, self.err = fn, init, None, None
        self.q = queue.Queue(maxsize=2)
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        if self.init is not None:
            self.init()  # (torch's current device is per thread)
        while True:
            item = self.q.get()
            try:
                if item is not None and self.err is None:
                    self.last = self.fn(item)
            except BaseException as e:  # noqa: BLE001 -- re-raised on the main thread by flush()
                self.err = e
            finally:
                self.q.task_done()
            if item is None:
                return

    def put(self, item):
        self.q.put(item)

    def flush(self):
        self.q.join()
        if self.err is not None:
            raise self.err
        return self.last

    def close(self):
        self.q.put(None)
        self.t.join()


class Shard:
    """one device's read block: resident synthetic input + a sketcher handle"""

    def __init__(self, F, S, device, first_read, n_reads, params, max_launch, profiling):
        self.device, self.first_read, self.n_reads = device, first_read, n_reads
        self.nbytes = n_reads * REC
        self.dg = F.DeviceBuffer(GENOME_LEN, device=device)
        self.dr = F.DeviceBuffer(self.nbytes + 64, device=device)
        S.synth_genome_device(self.dg, GENOME_LEN, SEED)
        S.synth_reads_device(self.dr, self.dg, GENOME_LEN, first_read, n_reads, READ_LEN, SEED, SUB_PPM, N_PPM)
        self.sk = params.create_sketcher(device=device, max_launch=max_launch)
        if profiling:
            self.sk.set_profiling(True)

    def step(self):
        sk = self.sk
        sk.reset()
        sk.set_stream_offset(self.first_read * REC)
        sk.push_device(self.dr.ptr, self.nbytes)
        kc, km, pos = sk.to_arrays()
        return kc, km, pos, sk.finish()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["c2", "c4", "c5"], default="c4",
                    help="c4 (default, every N): BASELINE configs[3], 50 Gbase in total; c2: configs[1], 10 Gbase per GPU; "
                         "c5: configs[4], batch of FASTA files through finch_sketch_files")
    ap.add_argument("--gbases", type=float, default=None, help="c2: Gbases per GPU (default 10); c4: Gbases in total (default 50)")
    ap.add_argument("--files", type=int, default=10000, help="c5: number of FASTA files")
    ap.add_argument("--c5-dir", default=None, help="c5: sketch the files g00000.fa ... already in this directory (the live counter passes of a "
                                                   "c5 run reuse their parent's files: a second set next to 40 GB of tmpfs is what the box's memory does not hold)")
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--cpu-sample-mbases", type=float, default=600.0)  # ~11 s of one host core
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the measurements reported under 'extras'")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="roofline.traffic from the committed PMC passes (profiles/pmc_summary.json) instead of counters collected by this run")
    ap.add_argument("--cpu-allcores-mbases", type=float, default=100.0,
                    help="Mbases per process for the extra all-cores CPU figure (0 = skip)")
    ap.add_argument("--max-launch", type=int, default=0)
    ap.add_argument("--backend", default="gloo",
                    help="torch.distributed backend when launched by torch.distributed.run.  The data path has no collective "
                         "(read blocks are independent; SURVEY 8e): the only traffic is the control-plane gather of one <= 40 KB "
                         "partial sketch per rank plus the timing reduction, which gloo carries as CPU tensors.  'nccl' (= RCCL) "
                         "moves the same gather onto the GPUs.")
    ap.add_argument("--share-gpu", action="store_true",
                    help="debug: map every rank / thread to device 0 to exercise the N > 1 flow on a 1-GPU box")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ  # torch.distributed.run: one process per GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else args.gpus
    if launched and world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    threads_mode = not launched and world > 1
    workload = args.workload
    gbases = args.gbases if args.gbases is not None else (10.0 if workload == "c2" else 50.0)

    import torch
    import finch_rs_amd as F
    from finch_rs_amd import sketch_schemes as S
    from finch_rs_amd import sharding as SH

    if not torch.cuda.is_available() or F.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libfinch_hip has no CPU path")
    if not args.share_gpu and not launched and F.device_count() < world:
        raise SystemExit("--gpus %d but only %d device(s) visible" % (world, F.device_count()))
    if launched and not args.share_gpu and local_rank >= F.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d device(s) visible: two ranks would share a GPU (--share-gpu to allow it)"
                         % (rank, local_rank, F.device_count()))
    if args.share_gpu:
        local_rank = 0
    # devices this PROCESS drives
    if threads_mode:
        my_devices = [0 if args.share_gpu else d for d in range(world)]
        my_ranks = list(range(world))
    else:
        my_devices, my_ranks = [local_rank], [rank]
    torch.cuda.set_device(my_devices[0])
    dist = None
    if launched and world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo" and os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # one node: keep gloo off hostname resolution (the container's hostname may not resolve)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    gather_device = "cuda" if args.backend == "nccl" else None

    def barrier():
        if dist is not None:
            dist.barrier()
        for d in sorted(set(my_devices)):
            torch.cuda.synchronize(d)

    if workload == "c5":
        return run_c5(args, F, S, dist, barrier, rank, world, my_devices, launched, gather_device)

    # ---- resident synthetic input (not timed) ----
    params = F.SketchParams.mash(args.n, args.n, True, args.k, 0)
    if workload == "c2":
        per = int(np.ceil(gbases * 1e9 / READ_LEN))  # per GPU
        total_reads = world * per
        bounds = {r: (r * per, (r + 1) * per) for r in my_ranks}
    else:
        total_reads = int(np.ceil(gbases * 1e9 / READ_LEN))  # in total: contiguous read blocks, one per GPU
        bounds = {r: SH.shard_bounds(total_reads, r, world) for r in my_ranks}
    shards = [Shard(F, S, d, bounds[r][0], bounds[r][1] - bounds[r][0], params, args.max_launch, profiling=True)
              for r, d in zip(my_ranks, my_devices)]

    # The merge of a step's partial sketches (<= n records per GPU, O(N n) on the host).  Launched by torch.distributed.run:
    # every rank ships its partial sketch to rank 0 as one small fixed-size tensor, on a thread of its own (MergePipe), so
    # the merge of step i runs behind the sketching of step i + 1.  One process: one fh_sketch_device_blocks call per step --
    # the library's own team of threads (one per device) and its own merge.
    kernel_acc = [0.0, 0, 0]  # ms, launches, positions of device 0's sketch launches
    rank_ms = {r: 0.0 for r in my_ranks}  # every rank's own sketch-kernel time (a slow device shows in the first real curve)

    def run_steps(n_steps, timed):
        if threads_mode:
            # ONE library call per step (fh_sketch_device_blocks): every device's reset / push / finish runs on a library thread
            # of its own and the partial sketches are merged on the calling thread inside the call -- no Python in the
            # per-device loop, so the GIL cannot serialise eight 10 ms steps
            last = None
            sks = [sh.sk for sh in shards]
            ptrs, lens = [sh.dr.ptr for sh in shards], [sh.nbytes for sh in shards]
            offs = [sh.first_read * REC for sh in shards]
            for _ in range(n_steps):
                SH.sketch_device_blocks(sks, ptrs, lens, offs)
                if timed:
                    for r, sh in zip(my_ranks, shards):
                        ms, nl, npos = sh.sk.kernel_time()
                        rank_ms[r] += ms
                        if r == 0:
                            kernel_acc[0] += ms; kernel_acc[1] += nl; kernel_acc[2] += npos
                kc, km, pos = shards[0].sk.to_arrays()
                last = (kc, km, pos, shards[0].sk.finish()[1])
            return last
        last = None
        for _ in range(n_steps):
            part = shards[0].step()
            if timed:
                ms, nl, npos = shards[0].sk.kernel_time()
                rank_ms[rank] += ms
                if rank == 0:
                    kernel_acc[0] += ms; kernel_acc[1] += nl; kernel_acc[2] += npos
            if merge is not None:
                merge.put(part)
            else:
                last = part
        return merge.flush() if merge is not None else last

    merge = None
    if dist is not None:
        merge = MergePipe(lambda part: SH.gather_and_merge(dist, params, part, args.n, device=gather_device),
                          init=lambda: torch.cuda.set_device(my_devices[0]))
    run_steps(args.warmup, False)
    # kernel-time accounting restarts with the timed region (reset() zeroes it)
    barrier()
    t0 = time.perf_counter()
    gathered = run_steps(args.steps, True)  # returns when every step's merge is done: inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, kernel_launches, kernel_pos = kernel_acc
    if merge is not None:
        merge.close()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        parts = [None] * world
        dist.all_gather_object(parts, rank_ms)
        for d in parts:
            rank_ms.update(d)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return 0

    n_reads = shards[0].n_reads
    dr, dg = shards[0].dr, shards[0].dg
    value = total_reads * READ_LEN * args.steps / elapsed
    # dominant kernel: k2_sketch.  Algorithmic bytes = 1 byte per k-mer start position it covers
    # (= 151/150 B per base for 150 bp reads), measured with HIP events on the
    # library's own stream around every launch (rank 0 / device 0).
    achieved = (kernel_pos / 1e9) / (kernel_ms / 1e3) if kernel_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "kernel": ("k2_sketch_seg<%d> (segments of %d positions: the reads' stride)" % (args.k, shards[0].sk.debug_segments()[2])
                           if shards[0].sk.debug_segments()[0] else "k2_sketch<%d>" % args.k), "launches": kernel_launches,
                "kernel_ms_per_pass": round(kernel_ms / max(args.steps, 1), 4),
                "avg_launch_ms": round(kernel_ms / max(kernel_launches, 1), 4),
                "alg_bytes_per_launch": int(kernel_pos / max(kernel_launches, 1)),
                "binding_resource": "VALU issue (integer hashing: murmur3's 64-bit multiplies and mixes per k-mer; DESIGN.md 3.1), "
                                    "not HBM -- see pmc"}
    # counter-derived figures come from the committed rocprofv3 PMC passes over exactly this command (profiles/README.md says
    # how each was collected); they are attached to the workload they were measured on and to no other
    is_default = (workload, gbases, args.k, args.n, world) == ("c4", 50.0, 21, 1000, 1)
    std_size = (workload, gbases) in (("c2", 10.0), ("c4", 50.0))
    pmc = _pmc_derived("%s_k%d_n%d" % (workload, args.k, args.n)) if (std_size and world == 1) else None
    if pmc:
        # HBM bytes the counters saw per algorithmic byte, applied to this run's launch size
        roofline["traffic"] = int(pmc["hbm_bytes_per_position"] * roofline["alg_bytes_per_launch"])
        roofline["traffic_source"] = "committed: " + str(pmc.get("source"))
        roofline["pmc"] = {k: pmc.get(k) for k in ("kernel", "valu_per_wave_iter", "cycles_per_wave_iter", "cycles_per_valu_inst", "valu_issue_model",
                                                   "lds_active_per_wave_iter", "lds_bank_conflict_per_wave_iter",
                                                   "hbm_bytes_per_position", "source")}

    # ... unless this run can collect them itself: the sketch launches' HBM bytes from PMC counters taken NOW (two profiled child
    # runs of this command, ~10 s each; not in the timed region), so that the traffic figure is this box's and this library's
    if world == 1 and std_size and not args.no_live_pmc:
        live = _live_pmc(["--workload", workload, "--gbases", repr(gbases), "--k", str(args.k), "--n", str(args.n)] +
                         (["--max-launch", str(args.max_launch)] if args.max_launch else []))
        if live:
            roofline["traffic"] = int(live["hbm_bytes_per_position"] * roofline["alg_bytes_per_launch"])
            roofline["traffic_source"] = live["source"]
            # the per-wave-iteration block is THIS run's (this box, this library); what the committed passes said about the same
            # workload stays next to it under its own name, with the file it came from
            if "pmc" in roofline:
                roofline["pmc_committed"] = roofline.pop("pmc")
            roofline["pmc"] = {k: live.get(k) for k in ("hbm_bytes_per_position", "fetch_size_kb", "write_size_kb", "valu_per_wave_iter",
                                                        "cycles_per_wave_iter", "cycles_per_valu_inst", "lds_active_per_wave_iter",
                                                        "lds_bank_conflict_per_wave_iter", "wait_any_frac_of_wave_cycles", "positions",
                                                        "dispatches", "seconds")}
            roofline["pmc"]["source"] = "live (this run)"
        elif LIVE_PMC_WHY:
            roofline["traffic_source"] = (roofline.get("traffic_source") or "") + " [live collection failed: %s]" % LIVE_PMC_WHY

    # measured streaming-read peak of this box next to the spec peak; not in the timed region
    try:
        roofline["measured_stream_read_GBps"] = round(S.measure_read_bandwidth(dr, min(shards[0].nbytes, 4 << 30) // 16 * 16), 1)
    except Exception:
        roofline["measured_stream_read_GBps"] = None

    cpu = None
    cpu_all = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O  # the checker, timed as the reported CPU baseline ("port")
        flags = O.use_native()  # compiled on this box with -march=native
        ns = min(n_reads, int(args.cpu_sample_mbases * 1e6 / READ_LEN))
        sample = dr.download(ns * REC)
        ora = O.OracleSketcher(O.MASH, args.n, args.k, 0)
        c0 = time.perf_counter()
        ora.process_packed(sample, 0)
        ct = time.perf_counter() - c0
        cpu = {"value": round(ns * READ_LEN / ct, 1), "unit": "bases/s", "cores": 1, "kind": "port",
               "sample": "first %d reads (%.0f Mbases) of the same stream; single thread = the reference's "
                         "behaviour for a single input file (rayon parallelises over files only); gcc %s"
                         % (ns, ns * READ_LEN / 1e6, flags)}
        # extra, NOT the reference's behaviour (it runs one input file on one core): the same oracle on read-block
        # shards of the stream, one process per host core, partial sketches merged afterwards
        if args.cpu_allcores_mbases > 0:
            import multiprocessing as mp
            ncpu = max(1, min(_usable_cpus(), 256))
            per = min(n_reads // ncpu, int(args.cpu_allcores_mbases * 1e6 / READ_LEN), int(3e9 / READ_LEN) // ncpu)
            if per > 0:
                _SHARED["big"] = dr.download(ncpu * per * REC)
                jobs = [(i, per * REC, args.n, args.k) for i in range(ncpu)]
                with mp.get_context("fork").Pool(ncpu) as p2:
                    p2.map(_oracle_shard_job, [(0, 151 * 64, args.n, args.k)] * ncpu, chunksize=1)  # start the workers
                    c0 = time.perf_counter()
                    p2.map(_oracle_shard_job, jobs, chunksize=1)
                    ct = time.perf_counter() - c0
                _SHARED.clear()
                cpu_all = {"value": round(ncpu * per * READ_LEN / ct, 1), "unit": "bases/s", "cores": ncpu, "kind": "port",
                           "sample": "%d read-block shards of %d reads, one oracle process per hardware thread (not what the "
                                     "reference does for a single file)" % (ncpu, per)}

    extras = None
    if world == 1 and not args.no_extras and is_default:
        global LIVE_PMC_EXTRAS
        LIVE_PMC_EXTRAS = not args.no_live_pmc
        extras = measure_extras(F, S, dr, dg, min(n_reads, int(np.ceil(10e9 / READ_LEN))), my_devices[0])

    drive = ("one process per GPU (torch.distributed.run, %s)" % args.backend if launched and world > 1 else
             "one process, one host thread + handle per GPU" if threads_mode else "one process, one GPU")
    if workload == "c2":
        wl = ("%.1f Gbase synthetic 150 bp reads per GPU (%s), mash k=%d n=%d seed 0, input resident in HBM as packed stream"
              % (gbases, "BASELINE configs[1]" if (gbases, args.k, args.n) == (10.0, 21, 1000) else "configs[1] generator, non-default "
                 "size/sketch", args.k, args.n))
        gkey = "c2_k%d_n%d" % (args.k, args.n) if (gbases, world) == (10.0, 1) else None
    else:
        wl = ("%.1f Gbase synthetic 150 bp reads in total (%s), split into %d contiguous read block%s (one per GPU), mash k=%d "
              "n=%d seed 0, blocks resident in HBM as packed stream, partial sketches merged on the host"
              % (gbases, "BASELINE configs[3]" if (gbases, args.k, args.n) == (50.0, 21, 1000) else "configs[3] generator, non-default "
                 "size/sketch", world, "" if world == 1 else "s", args.k, args.n))
        gkey = "c4_k%d_n%d" % (args.k, args.n) if gbases == 50.0 else None  # the same sketch whatever N
    # fingerprint of the final (merged) sketch: N GPUs over their read blocks must give what the oracle gives on the union
    check, ok = check_golden(fingerprint(gathered[0], gathered[1], gathered[3]), gkey) if gkey else \
        (dict(fingerprint(gathered[0], gathered[1], gathered[3]), golden=None, matches_golden=None), True)
    out = {
        "metric": "bases/sec sketched (k=%d, n=%d)" % (args.k, args.n),
        "value": round(value, 1), "unit": "bases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak" if workload == "c2" else "strong",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": wl, "reads_per_gpu": n_reads, "reads_total": total_reads,
                   "parallelism": "read-block sharding x%d, host merge; %s" % (world, drive)},
        "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_allcores": cpu_all,
        "per_rank": [{"rank": r, "reads": (total_reads // world if workload == "c2" else
                                           SH.shard_bounds(total_reads, r, world)[1] - SH.shard_bounds(total_reads, r, world)[0]),
                      "kernel_ms_per_pass": round(rank_ms[r] / max(args.steps, 1), 4)} for r in sorted(rank_ms)],
        "sketch_check": check,
    }
    if extras is not None:
        out["extras"] = extras
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0 if ok else 3


def run_c5(args, F, S, dist, barrier, rank, world, my_devices, launched, gather_device):
    """BASELINE configs[4]: a batch of FASTA files through finch_sketch_files (lib.rs:29-49), files mapped to GPUs"""
    import multiprocessing as mp
    import shutil
    import tempfile
    import torch
    from finch_rs_amd import host as H
    cands = [d for d in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(d)]
    base = max(cands, key=lambda d: shutil.disk_usage(d).free)
    free = shutil.disk_usage(base).free
    if base == "/dev/shm":  # tmpfs pages are RAM: leave room for the processes
        try:
            import psutil
            free = min(free, psutil.virtual_memory().available - (24 << 30))
        except Exception:
            pass
    nf = max(1, min(args.files, int(0.8 * free / (3.95e6 * 1.015))))  # mean of the log-uniform lengths + newlines
    d = os.path.join(base, "finch_bench_c5_%s" % os.environ.get("MASTER_PORT", str(os.getpid())))
    reuse = args.c5_dir is not None
    if reuse:
        d, nf = args.c5_dir, args.files
        base = os.path.dirname(d)
    try:
        if rank == 0 and not reuse:
            os.makedirs(d, exist_ok=True)
            with mp.get_context("fork").Pool(max(1, min(_usable_cpus(), 64))) as p2:
                made = p2.map(_write_fasta_job, [(d, i) for i in range(nf)], chunksize=8)
        barrier()
        paths = [os.path.join(d, "g%05d.fa" % i) for i in range(nf)]
        lens = [S.synth_fasta_length(i, SEED) for i in range(nf)]
        mine = list(range(rank, nf, world)) if launched else list(range(nf))
        params, filt = F.SketchParams.default(), H.FilterParams(None)
        res = None

        def step():
            nonlocal res
            res = H.sketch_files([paths[i] for i in mine], params, filt, devices=my_devices)
            assert len(res) == len(mine)

        for _ in range(args.warmup):
            step()
        barrier()
        H.debug_kernel_times(1)
        fb0 = H.debug_file_batch()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        k_ms, k_launches, k_pos = H.debug_kernel_times(0)
        fb1 = H.debug_file_batch()
        # fingerprint of the sample: files 0..255 (each rank contributes the ones it sketched)
        fx, tk, cs = 0, 0, 0
        for j, i in enumerate(mine):
            if i < C5_SAMPLE:
                sk = res.sketch(j)
                fx ^= int(np.bitwise_xor.reduce(sk.arrays[0]["hash"]))
                cs += int(sk.arrays[0]["count"].astype(np.uint64).sum())
                tk += int(sk.num_valid_kmers)
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            parts = [None] * world
            dist.all_gather_object(parts, (fx, tk, cs, k_ms, k_launches, k_pos))
            fx, tk, cs, k_ms, k_launches, k_pos = 0, 0, 0, 0.0, 0, 0
            for a, b, c, d1, d2, d3 in parts:
                fx ^= a; tk += b; cs += c; k_ms += d1; k_launches += d2; k_pos += d3
        if rank != 0:
            return 0
        tot = sum(lens)
        g = _golden("c5_files_0_255") if nf >= C5_SAMPLE else None
        fp = {"sample_files": min(nf, C5_SAMPLE), "hash_xor": fx, "count_sum": cs, "total_kmers": tk,
              "golden": "tests/golden/config_fingerprints.json[c5_files_0_255] (oracle, CPU)" if g else None,
              "matches_golden": None if g is None else all(g[k] == v for k, v in (("hash_xor", fx), ("count_sum", cs), ("total_kmers", tk)))}
        out = {"metric": "bases/sec sketched (k=21, n=1000)", "value": round(tot * args.steps / elapsed, 1), "unit": "bases/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": "BASELINE configs[4]: %d synthetic FASTA files (log-uniform 1-10 Mb, 70-column lines, %.2f Gbases, %s) "
                                      "through finch_sketch_files, library defaults (k=21 n=1000), files mapped to %d GPU%s%s"
                                      % (nf, tot / 1e9, base, world, "" if world == 1 else "s",
                                         "" if nf == args.files else "; %d asked for, cut to what %s holds" % (args.files, base)),
                          "files": nf, "files_per_s": round(nf * args.steps / elapsed, 1),
                          "parallelism": "file -> GPU mapping x%d (%s)" % (world, "one call per rank" if launched and world > 1 else "one call, devices=[0..%d]" % (world - 1))},
               "roofline": c5_roofline(k_ms, k_launches, k_pos, elapsed * 1e3, world, batch=(fb1[0] - fb0[0], fb1[1] - fb0[1]),
                                       live=None if (args.no_live_pmc or world != 1) else
                                       _live_pmc(["--workload", "c5", "--files", str(min(nf, 256)), "--c5-dir", d], timeout_s=240, kernels=("k2_batch", "k2_sketch")),
                                       h2d=_h2d_peak_gbs(my_devices[0]), text_bytes=sum(os.path.getsize(p) for p in paths) * args.steps),
               "cpu_baseline": None if args.no_cpu_baseline else c5_cpu_baseline(paths, lens), "sketch_check": fp}
        print(json.dumps(out), flush=True)
        return 0 if fp["matches_golden"] is not False else 3
    finally:
        res = None
        if dist is not None:
            dist.barrier()
        if rank == 0 and not reuse:
            shutil.rmtree(d, ignore_errors=True)
        if dist is not None:
            dist.destroy_process_group()


LIVE_PMC_EXTRAS = True  # main() clears it under --no-live-pmc


def measure_extras(F, S, dr, dg, n_reads, dev):
    """Measurements next to the headline one (same box, same run, OUTSIDE `value`): the other BASELINE configurations and the
    end-to-end rates, on the first `n_reads` reads (10 Gbase = configs[1]'s read set) of the resident stream.  Every entry says
    what it timed; a failing entry reports its error instead of taking the run down."""
    from finch_rs_amd import host as H
    bases = n_reads * READ_LEN
    ex = {}

    def resident(k, n, steps, warmup=1, gkey=None):
        """passes over the resident reads with a fresh sketcher; -> per-pass ms (best of `steps`), kernel GB/s, launches per pass"""
        p = F.SketchParams.mash(n, n, True, k, 0)
        s = p.create_sketcher(device=dev)
        s.set_profiling(True)
        best = 1e30
        kms = kl = kp = 0
        arrs, tk = None, 0
        for it in range(warmup + steps):
            t0 = time.perf_counter()
            s.reset()
            s.push_device(dr.ptr, n_reads * REC)
            arrs = s.to_arrays(out=arrs)  # (the caller's buffers are reused from the second pass on)
            tk = s.finish()[1]
            t1 = time.perf_counter()
            ms, nl, npos = s.kernel_time()
            if it >= warmup:
                best = min(best, t1 - t0)
                kms += ms; kl += nl; kp += npos
        dbg = s.debug_counters()
        s.close()
        r = {"ms_per_pass": round(best * 1e3, 3), "gbases_per_s": round(bases / best / 1e9, 2),
             "kernel_GBps": round(kp / 1e9 / (kms / 1e3), 2) if kms else None, "kernel_launches_per_pass": kl / max(steps, 1),
             "roofline_frac": round(kp / 1e9 / (kms / 1e3) / HBM_PEAK_GBS, 5) if kms else None, "big_prunes": dbg["big_prunes"]}
        if gkey and bases == 66666667 * READ_LEN:
            r["sketch_check"] = check_golden(fingerprint(arrs[0], arrs[1], tk), gkey)[0]
        return r

    def guarded(name, fn):
        try:
            ex[name] = fn()
        except Exception as e:  # noqa: BLE001 -- an extra must not take the headline number down
            ex[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    # -- BASELINE configs[1]: the first 10 Gbase of the stream on this one GPU --
    def c2():
        r = resident(21, 1000, 5, gkey="c2_k21_n1000")
        r["what"] = "BASELINE configs[1]: 10 Gbase (the first 66 666 667 reads of the stream), mash k=21 n=1000 (what `--workload c2` times)"
        r["pmc"] = _pmc_derived("c2_k21_n1000")
        return r
    guarded("c2_10gbase_k21_n1000", c2)

    # -- the same 10 Gbase, other sketches --
    def k31():
        r = resident(31, 1000, 3, gkey="c2_k31_n1000")
        r["what"] = ("configs[1]'s stream, mash k=31 n=1000 (at the VALU-issue AND the LDS-pipe ceiling: 78 VALU instructions and 8 table "
                     "lookups per position, DESIGN.md 5)")
        r["pmc"] = _pmc_derived("c2_k31_n1000")
        return r
    guarded("k31_n1000", k31)

    def k33():
        r = resident(33, 1000, 3)
        r["what"] = "configs[1]'s stream, mash k=33 n=1000: two-word k-mers (fh_k2w.hip)"
        r["pmc"] = _pmc_derived("c2_k33_n1000")
        return r
    guarded("k33_n1000", k33)

    def n200k():
        r = resident(21, 200_000, 3)
        r["what"] = "configs[1]'s stream, mash k=21 kmers_to_sketch=200000 (the CLI's default 200-fold oversketch, cli.rs:187-192)"
        return r
    guarded("k21_n200000", n200k)

    def c3():
        # the sketch as arrays (to_vec), best of three passes ...
        r = resident(31, 2_000_000, 3)
        # ... and the whole of configs[2]: sketch -> to_vec -> strand / error / abundance filters -> truncate to 10 000 hashes, on
        # the host in C++ (finch_sketch_from_sketcher = the tail of sketch_stream, lib.rs:70-93)
        pp = F.SketchParams.mash(2_000_000, 10_000, False, 31, 0)
        filt = H.FilterParams(True, (None, None), 0.31, 0.1)
        s = pp.create_sketcher(device=dev)
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            s.reset()
            s.push_device(dr.ptr, n_reads * REC)
            res = H.sketch_from_sketcher(s, "c3", bases, 2, pp, filt)
            best = min(best, time.perf_counter() - t0)
            assert H.lib().finch_sketch_n_hashes(res._p, 0) == 10_000
        sk0 = res.sketch(0)  # the filtered 10 000-hash sketch against the oracle's (sharded oracle -> its own filters)
        if bases == 66666667 * READ_LEN:
            r["sketch_check"] = check_golden(fingerprint(sk0.arrays[0], sk0.arrays[1], sk0.num_valid_kmers), "c3_k31_filtered")[0]
        s.close()
        r["ms_per_pass_with_host_filters"] = round(best * 1e3, 3)
        r["gbases_per_s_with_host_filters"] = round(bases / best / 1e9, 2)
        r["pmc"] = _pmc_derived("c3_k31_n2000000")
        r["what"] = ("BASELINE configs[2]: 10 Gbase, k=31, final 10000 hashes from kmers_to_sketch=2000000; with_host_filters adds strand "
                     "filter 0.1 + err filter 0.31 + truncate on the host (filter_counts, process_post_filter)")
        return r
    guarded("c3", c3)

    def fastq_text(ns):
        reads = dr.download(ns * REC).reshape(ns, REC)[:, :READ_LEN]
        w = 12 + READ_LEN + 3 + READ_LEN + 1  # "@r%09d\n" seq "\n+\n" qual "\n"
        txt = np.empty((ns, w), np.uint8)
        txt[:, 0], txt[:, 1] = ord("@"), ord("r")
        idx = np.arange(ns, dtype=np.int64)
        for d in range(9):
            txt[:, 10 - d] = 48 + (idx // 10 ** d) % 10
        txt[:, 11] = 10
        txt[:, 12:12 + READ_LEN] = reads
        txt[:, 12 + READ_LEN:15 + READ_LEN] = np.frombuffer(b"\n+\n", np.uint8)
        txt[:, 15 + READ_LEN:15 + 2 * READ_LEN] = ord("I")
        txt[:, w - 1] = 10
        return txt.reshape(-1)

    # -- end to end from FASTQ text in host memory (SURVEY 8d M1's separate line; PCIe and the device-side record
    #    splitting included; never `value`) --
    def e2e():
        ns = min(n_reads, 4_000_000)
        data = fastq_text(ns)
        p = F.SketchParams.mash(1000, 1000, True, 21, 0)
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            res = H.sketch_stream(data, "fastq", p, H.FilterParams(False), device=dev)
            best = min(best, time.perf_counter() - t0)
            assert H.lib().finch_sketch_seq_length(res._p, 0) == ns * READ_LEN
        return {"what": "finch_sketch_buffer on a %.2f GB plain FASTQ image in host memory (%d reads): staging into pinned buffers, "
                        "H2D, record splitting on the device (fh_push_fastq_text), sketch k=21 n=1000, finish"
                        % (data.size / 1e9, ns),
                "seconds": round(best, 4), "gbases_per_s": round(ns * READ_LEN / best / 1e9, 2),
                "text_GBps": round(data.size / best / 1e9, 2)}
    guarded("end_to_end_fastq_text", e2e)

    # -- the drop-in path at the trait level: sketch_stream's record loop calling process() --
    def trait_path():
        ns = min(n_reads, 4_000_000)
        reads = np.ascontiguousarray(dr.download(ns * REC))  # the records as a parser would hand them over: slices of one buffer
        offs = (np.arange(ns, dtype=np.uint64) * REC)
        lens = np.full(ns, READ_LEN, dtype=np.uint64)
        p = F.SketchParams.mash(1000, 1000, True, 21, 0)
        out = {"what": "sketch_stream's loop as the Rust binding runs it (INTEGRATION.md 2): one fh_process call per record of %d 150-base "
                       "records in host memory (one copy, blanks dropped on the way, into pinned staging; commits, H2D and kernels "
                       "behind it), then to_vec -- single thread, as the reference drives one file; block_*: the same bytes handed "
                       "over as 32 MB blocks of records + breakers through fh_push_block (the strip of a block runs on up to 8 threads)" % ns}
        s = p.create_sketcher(device=dev)
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            s.reset()
            s.process_records(reads, offs, lens)
            kc, km, _ = s.to_arrays()
            tk = s.finish()[1]
            best = min(best, time.perf_counter() - t0)
        out["seconds"] = round(best, 4)
        out["gbases_per_s"] = round(ns * READ_LEN / best / 1e9, 2)
        out["sequence_GBps"] = round(ns * READ_LEN / best / 1e9, 2)
        fp_rec = fingerprint(kc, km, tk)
        best = 1e30
        blk = 32 << 20
        for _ in range(3):
            t0 = time.perf_counter()
            s.reset()
            for o in range(0, reads.size, blk // REC * REC):
                s.push_block(reads[o:o + blk // REC * REC])
            kc, km, _ = s.to_arrays()
            tk = s.finish()[1]
            best = min(best, time.perf_counter() - t0)
        out["block_seconds"] = round(best, 4)
        out["block_sequence_GBps"] = round(ns * READ_LEN / best / 1e9, 2)
        out["both_forms_agree"] = fingerprint(kc, km, tk) == fp_rec
        s.close()
        return out
    guarded("trait_path", trait_path)

    # -- compressed input: the host inflates (fh_inflate.h), the device splits records and sketches --
    def gz():
        import shutil
        import struct
        import tempfile
        import zlib
        ns = min(n_reads, 1_000_000)
        raw = fastq_text(ns).tobytes()
        d = tempfile.mkdtemp(prefix="finch_bench_gz_")
        try:
            co = zlib.compressobj(1, zlib.DEFLATED, 31)
            gzp = os.path.join(d, "reads.fastq.gz")
            with open(gzp, "wb") as f:
                f.write(co.compress(raw) + co.flush())
            bgp = os.path.join(d, "reads.fastq.bgz")
            with open(bgp, "wb") as f:  # BGZF as bgzip writes it: independent members of <= 64 KiB with their size in the header
                for i in list(range(0, len(raw), 65280)) + [None]:
                    ch = b"" if i is None else raw[i:i + 65280]
                    c = zlib.compressobj(1, zlib.DEFLATED, -15)
                    body = c.compress(ch) + c.flush()
                    f.write(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(body) + 25) +
                            body + struct.pack("<II", zlib.crc32(ch), len(ch)))
            p = F.SketchParams.mash(1000, 1000, True, 21, 0)
            # (the sketchers the earlier lines parked fill the handle cache: without room there every call would allocate and
            # pin its buffers anew, which is not what a process that reads compressed files does)
            H._lib.load().fh_release_cached()
            out = {"what": "finch_sketch_files on one %.0f MB FASTQ (%d reads) compressed with zlib level 1: as a single gzip stream "
                           "(cut into chunks and inflated on the device, a wavefront per chunk; gzip_host_inflate: decoded by the call's read threads "
                           "together, fh_pargz.h) and as BGZF (members inflated on the device, one wavefront "
                           "each; bgzf_host_inflate: by the read threads instead); k=21 n=1000"
                           % (len(raw) / 1e6, ns)}
            gz_before = H.debug_device_gzip()
            for key, path, env in (("gzip", gzp, None), ("gzip_host_inflate", gzp, "0"), ("bgzf", bgp, None), ("bgzf_host_inflate", bgp, "0")):
                if env is not None:
                    F.debug_set(device_inflate=env)
                try:
                    best = 1e30
                    for _ in range(3):
                        t0 = time.perf_counter()
                        res = H.sketch_files([path], p, H.FilterParams(False), devices=[dev])
                        best = min(best, time.perf_counter() - t0)
                        assert H.lib().finch_sketch_seq_length(res._p, 0) == ns * READ_LEN
                finally:
                    F.debug_set(device_inflate=None)
                out[key + "_gbases_per_s"] = round(ns * READ_LEN / best / 1e9, 3)
                out[key + "_text_GBps"] = round(len(raw) / best / 1e9, 3)
            out["bgzf_inflated_on_device"] = H.debug_device_inflate()[0] > 0
            gz_after = H.debug_device_gzip()
            out["gzip_inflated_on_device"] = gz_after[0] - gz_before[0] >= 3 and gz_after[1] == gz_before[1]
            out["gzip_feed_timeouts"] = int(H._lib.load().fh_debug_gzip_feed_timeouts())  # (0, or the line above measured a fallback)
            return out
        finally:
            shutil.rmtree(d, ignore_errors=True)
    guarded("compressed_fastq", gz)

    # -- configs[4]'s shape on one GPU: a batch of FASTA files through ONE finch_sketch_files call (`--workload c5` times the
    #    full 10 000) --
    def c5():
        import multiprocessing as mp
        import shutil
        import tempfile
        nf = 1024
        cands = [d for d in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(d)]
        base = max(cands, key=lambda d: shutil.disk_usage(d).free)
        if shutil.disk_usage(base).free < 8e9:
            return {"error": "no room for 4 GB of FASTA files under %s" % base}
        d = tempfile.mkdtemp(prefix="finch_bench_c5_", dir=base)
        try:
            with mp.get_context("fork").Pool(max(1, min(_usable_cpus(), 32))) as pool:
                made = pool.map(_write_fasta_job, [(d, i) for i in range(nf)], chunksize=8)
            paths = [m[0] for m in made]
            tot = sum(m[1] for m in made)
            best = 1e30
            # (the sketchers the earlier lines parked fill the handle cache -- option pool_bytes: with them there, part of this
            # batch's sixteen worker handles would be allocated and pinned anew in every call, which is not what a process that
            # sketches batches does)
            H._lib.load().fh_release_cached()
            kt = (0.0, 0, 0)
            fb = (0, 0)
            res = None
            for _ in range(4):
                H.debug_kernel_times(1)
                fb0 = H.debug_file_batch()
                res = None  # (the previous call's sketches are dropped here, not inside the next call's time)
                t0 = time.perf_counter()
                res = H.sketch_files(paths, F.SketchParams.default(), H.FilterParams(None), devices=[dev])
                dt = time.perf_counter() - t0
                fb1 = H.debug_file_batch()
                if dt < best:
                    best, kt, fb = dt, H.debug_kernel_times(0), (fb1[0] - fb0[0], fb1[1] - fb0[1])
                assert len(res) == nf
            H.debug_kernel_times(0)
            return {"what": "ONE finch_sketch_files call over %d synthetic FASTA files (log-uniform 1-10 Mb, 70-column lines, "
                            "%.2f Gbases, page cache / tmpfs), library defaults (k=21 n=1000, up to 16 worker threads per GPU)" % (nf, tot / 1e9),
                    "seconds": round(best, 4), "files_per_s": round(nf / best, 1), "gbases_per_s": round(tot / best / 1e9, 2),
                    "roofline": c5_roofline(kt[0], kt[1], kt[2], best * 1e3, 1, batch=fb, h2d=_h2d_peak_gbs(dev), text_bytes=sum(os.path.getsize(p) for p in paths),
                                            live=_live_pmc(["--workload", "c5", "--files", "256"], kernels=("k2_batch", "k2_sketch"))
                                            if LIVE_PMC_EXTRAS else None),
                    "cpu_baseline": c5_cpu_baseline(paths, [m[1] for m in made])}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    guarded("c5_batch_1gpu", c5)
    return ex


if __name__ == "__main__":
    sys.exit(main())
