#!/usr/bin/env python3
"""bench.py -- bases/sec sketched (k=21, n=1000) on N x MI355X, with the kernel's HBM roofline
fraction and the CPU baseline timed beside it (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4] [--gbases G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workloads (synthetic 150 bp FASTQ-shaped read sets, SURVEY.md 8d M4, already resident in HBM as the packed
sequence stream -- 150 bases + 1 breaker byte per read -- when the timed region starts; Mash sketch k=21,
kmers_to_sketch=1000, seed 0):
  c2  BASELINE.json configs[1]: 10 Gbase on one GPU.  The default for N = 1 (the configuration the metric is quoted on).
      With N > 1 every rank gets its own 10 Gbase (weak scaling).
  c4  BASELINE.json configs[3]: 50 Gbase IN TOTAL, the reads split into N contiguous read blocks (shard_bounds), one
      per GPU, partial sketches merged on the host of rank 0 -- strong scaling.  The default for N > 1.
A step = one full pass: reset, sketch every base of the rank's read block, finish (bottom-n select, copy-out of the
<= 1000 records to the host) and -- for N > 1 -- the host-side merge of the partial sketches on rank 0 (no data-path
collective; read blocks are independent).  value = total bases of all ranks * K / max-over-ranks time.

After the timed region rank 0 of an N = 1 run also measures, OUTSIDE `value` (key "extras"): the same stream at k=31,
BASELINE's configs[2] sketch (k=31, 2 M hashes, host filters), the CLI-default oversketch (n=200 000), configs[3] on one
GPU (50 Gbase), the end-to-end rate from FASTQ text in host memory and a batch of FASTA files through
finch_sketch_files (configs[4]'s shape on one GPU).  --no-extras skips them.
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
SUB_PPM, N_PPM = 10_000, 500
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    return x ^ (x >> 31)


def _write_fasta_job(job):
    """one synthetic genome as a 70-column FASTA file (log-uniform 1..10 Mb from the per-file seed)"""
    from finch_rs_amd import sketch_schemes as S
    d, i = job
    u = (_splitmix64(SEED + 7919 * i) >> 11) / float(1 << 53)
    L = int(1e6 * 10.0 ** u)
    g = S.synth_genome_host(L, SEED + 1000003 * (i + 1))
    rows = (L + 69) // 70
    a = np.full((rows, 71), 10, np.uint8)
    gp = np.zeros(rows * 70, np.uint8)
    gp[:L] = g
    a[:, :70] = gp.reshape(rows, 70)
    path = os.path.join(d, "g%05d.fa" % i)
    with open(path, "wb") as f:
        f.write(b">genome_%05d len=%d\n" % (i, L))
        f.write(a.reshape(-1)[:(rows - 1) * 71 + L - (rows - 1) * 70].tobytes() + b"\n")
    return path, L


def _pmc_derived(key):
    """what the committed rocprofv3 PMC passes say about the kernel's binding resource (profiles/pmc_summary.json, written
    by tools/pmc_summary.py from the counter files named there); None if no set was collected for this workload"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json"))).get(key)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["c2", "c4"], default=None, help="default: c2 for --gpus 1, c4 otherwise")
    ap.add_argument("--gbases", type=float, default=None, help="c2: Gbases per GPU (default 10); c4: Gbases in total (default 50)")
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--cpu-sample-mbases", type=float, default=450.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the measurements reported under 'extras'")
    ap.add_argument("--cpu-allcores-mbases", type=float, default=100.0,
                    help="Mbases per process for the extra all-cores CPU figure (0 = skip)")
    ap.add_argument("--max-launch", type=int, default=0)
    ap.add_argument("--backend", default="gloo",
                    help="torch.distributed backend for N>1.  The data path has no collective (read blocks are "
                         "independent; SURVEY 8e): the only traffic is the control-plane gather of one <= 40 KB partial "
                         "sketch per rank plus the timing reduction, which gloo carries as CPU tensors.  'nccl' (= RCCL) "
                         "moves the same gather onto the GPUs.")
    ap.add_argument("--share-gpu", action="store_true",
                    help="debug: map every rank to cuda:0 (with --backend gloo) to exercise the N>1 flow on a 1-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    workload = args.workload or ("c2" if world == 1 else "c4")
    gbases = args.gbases if args.gbases is not None else (10.0 if workload == "c2" else 50.0)

    import torch
    import finch_rs_amd as F
    from finch_rs_amd import sketch_schemes as S
    from finch_rs_amd import sharding as SH

    if not torch.cuda.is_available() or F.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libfinch_hip has no CPU path")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
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
        torch.cuda.synchronize()

    # ---- resident synthetic input (not timed) ----
    rec = READ_LEN + 1
    if workload == "c2":
        n_reads = int(np.ceil(gbases * 1e9 / READ_LEN))  # per GPU
        first_read = rank * n_reads
        total_reads = world * n_reads
    else:
        total_reads = int(np.ceil(gbases * 1e9 / READ_LEN))  # in total: contiguous read blocks, one per rank
        first_read, hi = SH.shard_bounds(total_reads, rank, world)
        n_reads = hi - first_read
    nbytes = n_reads * rec
    dg = F.DeviceBuffer(GENOME_LEN, device=local_rank)
    dr = F.DeviceBuffer(nbytes + 64, device=local_rank)
    S.synth_genome_device(dg, GENOME_LEN, SEED)
    S.synth_reads_device(dr, dg, GENOME_LEN, first_read, n_reads, READ_LEN, SEED, SUB_PPM, N_PPM)

    params = F.SketchParams.mash(args.n, args.n, True, args.k, 0)
    sk = params.create_sketcher(device=local_rank, max_launch=args.max_launch)
    sk.set_profiling(True)

    gathered = None

    def step():
        nonlocal gathered
        sk.reset()
        sk.set_stream_offset(first_read * rec)
        sk.push_device(dr.ptr, nbytes)
        kc, km, pos = sk.to_arrays()
        tk = sk.finish()[1]
        if dist is not None:
            # partial sketches are <= n records: ship them to rank 0 (one small fixed-size tensor per rank)
            # and merge on the host, O(N*n) -- finch_rs_amd/sharding.py
            merged = SH.gather_and_merge(dist, params, (kc, km, pos, tk), args.n, device=gather_device)
            if rank == 0:
                gathered = merged[:3]
        else:
            gathered = (kc, km, pos)

    for _ in range(args.warmup):
        step()
    # kernel-time accounting restarts with the timed region (reset() zeroes it)
    barrier()
    t0 = time.perf_counter()
    kernel_ms, kernel_launches, kernel_pos = 0.0, 0, 0
    for _ in range(args.steps):
        step()
        ms, nl, npos = sk.kernel_time()
        kernel_ms += ms; kernel_launches += nl; kernel_pos += npos
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = total_reads * READ_LEN * args.steps / elapsed
    # dominant kernel: k2_sketch.  Algorithmic bytes = 1 byte per k-mer start position it covers
    # (= 151/150 B per base for 150 bp reads), measured with HIP events on the
    # library's own stream around every launch (rank 0).
    achieved = (kernel_pos / 1e9) / (kernel_ms / 1e3) if kernel_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "kernel": "k2_sketch<%d>" % args.k, "launches": kernel_launches,
                "avg_launch_ms": round(kernel_ms / max(kernel_launches, 1), 4),
                "alg_bytes_per_launch": int(kernel_pos / max(kernel_launches, 1)),
                "binding_resource": "VALU issue (integer hashing: murmur3's 64-bit multiplies and mixes per k-mer; DESIGN.md 3.1), "
                                    "not HBM -- see pmc"}
    # counter-derived figures come from the committed rocprofv3 PMC passes over exactly this command (profiles/README.md says
    # how each was collected); they are attached to the workload they were measured on and to no other
    is_default = (workload, gbases, args.k, args.n, world) == ("c2", 10.0, 21, 1000, 1)
    pmc = _pmc_derived("c2_k%d_n%d" % (args.k, args.n)) if (workload, gbases, world) == ("c2", 10.0, 1) else None
    if pmc:
        roofline["traffic"] = pmc.get("hbm_bytes_per_launch")
        roofline["pmc"] = {k: pmc.get(k) for k in ("valu_per_wave_iter", "valu_busy", "cycles_per_wave_iter", "cycles_per_valu_inst",
                                                   "lds_active_per_wave_iter", "lds_bank_conflict_per_wave_iter",
                                                   "hbm_bytes_per_position", "source")}
    elif is_default:
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            roofline["traffic"] = json.load(open(prof)).get("k2_hbm_bytes_per_launch")
        except Exception:
            pass

    # measured streaming-read peak of this box next to the spec peak; not in the timed region
    try:
        roofline["measured_stream_read_GBps"] = round(S.measure_read_bandwidth(dr, min(nbytes, 4 << 30) // 16 * 16), 1)
    except Exception:
        roofline["measured_stream_read_GBps"] = None

    cpu = None
    cpu_all = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O  # the checker, timed as the reported CPU baseline ("port")
        flags = O.use_native()  # compiled on this box with -march=native
        ns = min(n_reads, int(args.cpu_sample_mbases * 1e6 / READ_LEN))
        sample = dr.download(ns * rec)
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
                _SHARED["big"] = dr.download(ncpu * per * rec)
                jobs = [(i, per * rec, args.n, args.k) for i in range(ncpu)]
                with mp.get_context("fork").Pool(ncpu) as pool:
                    pool.map(_oracle_shard_job, [(0, 151 * 64, args.n, args.k)] * ncpu, chunksize=1)  # start the workers
                    c0 = time.perf_counter()
                    pool.map(_oracle_shard_job, jobs, chunksize=1)
                    ct = time.perf_counter() - c0
                _SHARED.clear()
                cpu_all = {"value": round(ncpu * per * READ_LEN / ct, 1), "unit": "bases/s", "cores": ncpu, "kind": "port",
                           "sample": "%d read-block shards of %d reads, one oracle process per hardware thread (not what the "
                                     "reference does for a single file)" % (ncpu, per)}

    extras = None
    if world == 1 and not args.no_extras and is_default:
        extras = measure_extras(F, S, dr, dg, n_reads, nbytes, local_rank)

    if workload == "c2":
        wl = ("%.1f Gbase synthetic 150 bp reads per GPU (%s), mash k=%d n=%d seed 0, input resident in HBM as packed stream"
              % (gbases, "configs[1]" if (gbases, args.k, args.n) == (10.0, 21, 1000) else "configs[1] generator, non-default "
                 "size/sketch", args.k, args.n))
    else:
        wl = ("%.1f Gbase synthetic 150 bp reads in total (%s), split into %d contiguous read blocks (one per GPU), mash k=%d "
              "n=%d seed 0, blocks resident in HBM as packed stream, partial sketches merged on the host"
              % (gbases, "configs[3]" if (gbases, args.k, args.n) == (50.0, 21, 1000) else "configs[3] generator, non-default "
                 "size/sketch", world, args.k, args.n))
    out = {
        "metric": "bases/sec sketched (k=%d, n=%d)" % (args.k, args.n),
        "value": round(value, 1), "unit": "bases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak" if workload == "c2" else "strong",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": wl, "reads_per_gpu": n_reads, "reads_total": total_reads,
                   "parallelism": "read-block sharding x%d, host merge" % world},
        "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_allcores": cpu_all,
        # fingerprint of the final (merged) sketch: N ranks over their read blocks must give what one rank gives on the union
        "sketch_check": {"n_hashes": int(len(gathered[0])), "min_hash": int(gathered[0]["hash"][0]) if len(gathered[0]) else None,
                         "max_hash": int(gathered[0]["hash"][-1]) if len(gathered[0]) else None,
                         "hash_xor": int(np.bitwise_xor.reduce(gathered[0]["hash"])) if len(gathered[0]) else 0,
                         "count_sum": int(gathered[0]["count"].astype(np.uint64).sum()),
                         "extra_sum": int(gathered[0]["extra_count"].astype(np.uint64).sum()),
                         "kmer_byte_sum": int(gathered[1].astype(np.uint64).sum())},
    }
    if extras is not None:
        out["extras"] = extras
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def measure_extras(F, S, dr, dg, n_reads, nbytes, dev):
    """Measurements next to the headline one (same box, same run, OUTSIDE `value`): the other BASELINE configurations and the
    end-to-end rates.  Every entry says what it timed; a failing entry reports its error instead of taking the run down."""
    from finch_rs_amd import host as H
    rec = READ_LEN + 1
    bases = n_reads * READ_LEN
    ex = {}

    def resident(k, n, steps, warmup=1, after=None, buf=None, reads=None):
        """passes over a resident stream with a fresh sketcher; -> per-pass ms (best of `steps`), kernel GB/s, launches per pass"""
        buf = buf or dr
        reads = reads or n_reads
        p = F.SketchParams.mash(n, n, True, k, 0)
        s = p.create_sketcher(device=dev)
        s.set_profiling(True)
        best, best_after = 1e30, 1e30
        kms = kl = kp = 0
        for it in range(warmup + steps + (1 if after is not None else 0)):
            t0 = time.perf_counter()
            s.reset()
            s.push_device(buf.ptr, reads * rec)
            arrs = s.to_arrays()
            tk = s.finish()[1]
            t1 = time.perf_counter()
            ms, nl, npos = s.kernel_time()
            if it >= warmup + steps:  # one more pass, with the host-side post-processing behind it
                after(p, arrs, tk)
                best_after = time.perf_counter() - t0
            elif it >= warmup:
                best = min(best, t1 - t0)
                kms += ms; kl += nl; kp += npos
            del arrs
        dbg = s.debug_counters()
        s.close()
        r = {"ms_per_pass": round(best * 1e3, 3), "gbases_per_s": round(reads * READ_LEN / best / 1e9, 2),
             "kernel_GBps": round(kp / 1e9 / (kms / 1e3), 2) if kms else None, "kernel_launches_per_pass": kl / max(steps, 1),
             "roofline_frac": round(kp / 1e9 / (kms / 1e3) / HBM_PEAK_GBS, 5) if kms else None, "big_prunes": dbg["big_prunes"]}
        if after is not None:
            r["ms_per_pass_with_host_filters"] = round(best_after * 1e3, 3)
            r["gbases_per_s_with_host_filters"] = round(reads * READ_LEN / best_after / 1e9, 2)
        return r

    def guarded(name, fn):
        try:
            ex[name] = fn()
        except Exception as e:  # noqa: BLE001 -- an extra must not take the headline number down
            ex[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    # -- the same 10 Gbase stream, other sketches --
    def k31():
        r = resident(31, 1000, 3)
        r["what"] = "configs[1]'s stream, mash k=31 n=1000 (bound by the LDS pipe: 6-8 table lookups per position)"
        r["pmc"] = _pmc_derived("c2_k31_n1000")
        return r
    guarded("k31_n1000", k31)

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
            s.push_device(dr.ptr, n_reads * rec)
            res = H.sketch_from_sketcher(s, "c3", bases, 2, pp, filt)
            best = min(best, time.perf_counter() - t0)
            assert H.lib().finch_sketch_n_hashes(res._p, 0) == 10_000
        s.close()
        r["ms_per_pass_with_host_filters"] = round(best * 1e3, 3)
        r["gbases_per_s_with_host_filters"] = round(bases / best / 1e9, 2)
        r["what"] = ("BASELINE configs[2]: 10 Gbase, k=31, final 10000 hashes from kmers_to_sketch=2000000; with_host_filters adds strand "
                     "filter 0.1 + err filter 0.31 + truncate on the host (filter_counts, process_post_filter)")
        return r
    guarded("c3", c3)

    # -- configs[3] on one GPU: the 50 Gbase stream resident --
    def c4():
        reads50 = int(np.ceil(50e9 / READ_LEN))
        d50 = F.DeviceBuffer(reads50 * rec + 64, device=dev)
        try:
            S.synth_reads_device(d50, dg, GENOME_LEN, 0, reads50, READ_LEN, SEED, SUB_PPM, N_PPM)
            r = resident(21, 1000, 2, buf=d50, reads=reads50)
        finally:
            d50.free()
        r["what"] = "BASELINE configs[3] on ONE GPU: 50 Gbase resident, mash k=21 n=1000 (what `--gpus 1 --workload c4` times)"
        return r
    guarded("c4_50gbase_1gpu", c4)

    # -- end to end from FASTQ text in host memory (SURVEY 8d M1's separate line; PCIe and the device-side record
    #    splitting included; never `value`) --
    def e2e():
        ns = min(n_reads, 4_000_000)
        reads = dr.download(ns * rec).reshape(ns, rec)[:, :READ_LEN]
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
        data = txt.reshape(-1)
        del txt
        p = F.SketchParams.mash(1000, 1000, True, 21, 0)
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            res = H.sketch_stream(data, "fastq", p, H.FilterParams(False), device=dev)
            best = min(best, time.perf_counter() - t0)
            assert H.lib().finch_sketch_seq_length(res._p, 0) == ns * READ_LEN
        return {"what": "finch_sketch_buffer on a %.2f GB plain FASTQ image in host memory (%d reads): copy into the pinned staging "
                        "buffer, H2D, record splitting on the device (fh_push_fastq_text), sketch k=21 n=1000, finish"
                        % (data.size / 1e9, ns),
                "seconds": round(best, 4), "gbases_per_s": round(ns * READ_LEN / best / 1e9, 2),
                "text_GBps": round(data.size / best / 1e9, 2)}
    guarded("end_to_end_fastq_text", e2e)

    # -- compressed input: the host inflates (fh_inflate.h), the device splits records and sketches --
    def gz():
        import shutil
        import struct
        import tempfile
        import zlib
        ns = min(n_reads, 1_000_000)
        reads = dr.download(ns * rec).reshape(ns, rec)[:, :READ_LEN]
        w = 12 + READ_LEN + 3 + READ_LEN + 1
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
        raw = txt.tobytes()
        del txt
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
                           "(decoded by the call's read threads together, fh_pargz.h) and as BGZF (members inflated on the device, one wavefront "
                           "each; bgzf_host_inflate: by the read threads instead); k=21 n=1000"
                           % (len(raw) / 1e6, ns)}
            for key, path, env in (("gzip", gzp, None), ("bgzf", bgp, None), ("bgzf_host_inflate", bgp, "0")):
                if env is not None:
                    os.environ["FINCH_DEVICE_INFLATE"] = env
                try:
                    best = 1e30
                    for _ in range(3):
                        t0 = time.perf_counter()
                        res = H.sketch_files([path], p, H.FilterParams(False), devices=[dev])
                        best = min(best, time.perf_counter() - t0)
                        assert H.lib().finch_sketch_seq_length(res._p, 0) == ns * READ_LEN
                finally:
                    os.environ.pop("FINCH_DEVICE_INFLATE", None)
                out[key + "_gbases_per_s"] = round(ns * READ_LEN / best / 1e9, 3)
                out[key + "_text_GBps"] = round(len(raw) / best / 1e9, 3)
            out["bgzf_inflated_on_device"] = H.debug_device_inflate()[0] > 0
            return out
        finally:
            shutil.rmtree(d, ignore_errors=True)
    guarded("compressed_fastq", gz)

    # -- configs[4]'s shape on one GPU: a batch of FASTA files through ONE finch_sketch_files call --
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
            for _ in range(3):
                t0 = time.perf_counter()
                res = H.sketch_files(paths, F.SketchParams.default(), H.FilterParams(None), devices=[dev])
                best = min(best, time.perf_counter() - t0)
                assert len(res) == nf
            return {"what": "ONE finch_sketch_files call over %d synthetic FASTA files (log-uniform 1-10 Mb, 70-column lines, "
                            "%.2f Gbases, page cache / tmpfs), library defaults (k=21 n=1000, 12 worker threads per GPU)" % (nf, tot / 1e9),
                    "seconds": round(best, 4), "files_per_s": round(nf / best, 1), "gbases_per_s": round(tot / best / 1e9, 2)}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    guarded("c5_batch_1gpu", c5)
    return ex


if __name__ == "__main__":
    main()
