#!/usr/bin/env python3
"""bench.py -- bases/sec sketched (k=21, n=1000) on N x MI355X, with the kernel's HBM roofline
fraction and the CPU baseline timed beside it (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--gbases G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): a synthetic 150 bp FASTQ-shaped read set of G Gbases PER GPU
(default 10), already resident in HBM as the packed sequence stream (150 bases + 1 breaker byte per
read) when the timed region starts; Mash sketch k=21, kmers_to_sketch=1000, seed 0.
A step = one full pass: reset, sketch every base of the rank's read block, finish (bottom-n select,
copy-out of the <=1000 records to the host) and -- for N>1 -- the host-side merge of the partial
sketches on rank 0 (no data-path collective; read blocks are independent).
Scaling is weak: every rank sketches its own G Gbases; value = N*G*1e9*K / max-over-ranks time.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gbases", type=float, default=10.0, help="Gbases per GPU")
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--cpu-sample-mbases", type=float, default=450.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    n_reads = int(np.ceil(args.gbases * 1e9 / READ_LEN))
    rec = READ_LEN + 1
    nbytes = n_reads * rec
    bases = n_reads * READ_LEN
    first_read = rank * n_reads
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

    value = world * bases * args.steps / elapsed
    # dominant kernel: k2_sketch.  Algorithmic bytes = 1 byte per k-mer start position it covers
    # (= 151/150 B per base for 150 bp reads), measured with HIP events on the
    # library's own stream around every launch (rank 0).
    achieved = (kernel_pos / 1e9) / (kernel_ms / 1e3) if kernel_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "kernel": "k2_sketch<%d>" % args.k, "launches": kernel_launches,
                "avg_launch_ms": round(kernel_ms / max(kernel_launches, 1), 4),
                "alg_bytes_per_launch": int(kernel_pos / max(kernel_launches, 1)),
                "note": "integer-ALU bound by construction (murmur3: four 64-bit multiplies + three mad-based key-word mixes per k-mer, VALU ~92% busy); see DESIGN.md 3.1"}
    prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    # the committed PMC figure was collected on exactly the default workload; do not attach it to another one
    if os.path.exists(prof) and (args.gbases, args.k, args.n, world) == (10.0, 21, 1000, 1):
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
        ns = min(n_reads, int(args.cpu_sample_mbases * 1e6 / READ_LEN))
        sample = dr.download(ns * rec)
        ora = O.OracleSketcher(O.MASH, args.n, args.k, 0)
        c0 = time.perf_counter()
        ora.process_packed(sample, 0)
        ct = time.perf_counter() - c0
        cpu = {"value": round(ns * READ_LEN / ct, 1), "unit": "bases/s", "cores": 1, "kind": "port",
               "sample": "first %d reads (%.0f Mbases) of the same stream; single thread = the reference's "
                         "behaviour for a single input file (rayon parallelises over files only)" % (ns, ns * READ_LEN / 1e6)}
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
    out = {
        "metric": "bases/sec sketched (k=%d, n=%d)" % (args.k, args.n),
        "value": round(value, 1), "unit": "bases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%.1f Gbase synthetic 150 bp reads per GPU (%s), mash k=%d n=%d seed 0, "
                               "input resident in HBM as packed stream"
                               % (args.gbases, "configs[1]" if (args.gbases, args.k, args.n) == (10.0, 21, 1000)
                                  else "configs[1] generator, non-default size/sketch", args.k, args.n),
                   "reads_per_gpu": n_reads, "parallelism": "read-block sharding x%d, host merge" % world},
        "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_allcores": cpu_all,
        # fingerprint of the final (merged) sketch: N ranks x G Gbase must give what one rank gives on N*G Gbase
        "sketch_check": {"n_hashes": int(len(gathered[0])), "min_hash": int(gathered[0]["hash"][0]) if len(gathered[0]) else None,
                         "max_hash": int(gathered[0]["hash"][-1]) if len(gathered[0]) else None,
                         "hash_xor": int(np.bitwise_xor.reduce(gathered[0]["hash"])) if len(gathered[0]) else 0,
                         "count_sum": int(gathered[0]["count"].astype(np.uint64).sum()),
                         "extra_sum": int(gathered[0]["extra_count"].astype(np.uint64).sum()),
                         "kmer_byte_sum": int(gathered[1].astype(np.uint64).sum())},
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
