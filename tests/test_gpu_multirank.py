"""bench.py's N > 1 flows on a 1-GPU box (--share-gpu maps every rank / thread to device 0):
  * launched plainly (`python bench.py --gpus N`): one process, one fh_sketch_device_blocks call per step (one library thread +
    sketcher handle per device, merge inside the call);
  * launched by torch.distributed.run: one process per rank, rank 0 gathers and merges.
Each rank sketches its own read block; the merged sketch of N ranks must be the sketch one rank computes on the union (same read
indices): SURVEY 8e through the real launchers, timing protocol and JSON contract included -- and, at BASELINE configs[3]'s
full size, it must be the sketch the ORACLE computed on the CPU (tests/golden/config_fingerprints.json)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP = ("n_hashes", "min_hash", "max_hash", "hash_xor", "count_sum", "extra_sum", "kmer_byte_sum", "total_kmers")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(args, world=1, launcher=True, rc=0):
    env = dict(os.environ)
    if world > 1 and launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", str(world), "--share-gpu"]
    else:
        cmd = [sys.executable, "bench.py"] + (["--gpus", str(world), "--share-gpu"] if world > 1 else [])
        for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):  # a plain launch: nothing of a launcher in the environment
            env.pop(v, None)
    r = subprocess.run(cmd + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert r.returncode == rc, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def _fp(d):
    return {k: d["sketch_check"][k] for k in FP}


@pytest.mark.parametrize("launcher", [False, True], ids=["threads", "torchrun"])
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_times_block_equals_one_rank_on_everything(world, launcher):
    """--workload c2 (weak scaling): every rank its own G Gbase"""
    g = 0.06
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--workload", "c2"]
    multi = _bench(["--gbases", str(g)] + common, world, launcher)
    single = _bench(["--gbases", str(g * world)] + common, 1)
    assert multi["n_gpus"] == world and multi["steps"] == 2 and multi["scaling"] == "weak"
    assert multi["unit"] == "bases/s" and multi["value"] > 0 and multi["roofline"]["bound"] == "hbm"
    assert multi["sketch_check"]["n_hashes"] == 1000 and multi["sketch_check"]["matches_golden"] is None
    assert _fp(multi) == _fp(single)


@pytest.mark.parametrize("launcher", [False, True], ids=["threads", "torchrun"])
@pytest.mark.parametrize("world", [2, 3])
def test_c4_workload_splits_one_read_set_into_read_blocks(world, launcher):
    """the default workload is BASELINE configs[3]'s shape (strong scaling) for every N: G Gbase IN TOTAL, rank r takes the
    read block shard_bounds(R, r, N); the merged sketch must be the one a single rank computes on all R reads, and `value`
    counts the total once."""
    g = 0.2
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--gbases", str(g)]
    multi = _bench(common, world, launcher)  # (--workload defaults to c4 for every N)
    single = _bench(common, 1)
    assert multi["n_gpus"] == world and multi["scaling"] == "strong" and single["scaling"] == "strong"
    assert "configs[3] generator" in multi["config"]["workload"] and "%d contiguous read blocks" % world in multi["config"]["workload"]
    assert ("one host thread + handle per GPU" in multi["config"]["parallelism"]) == (not launcher)
    assert multi["config"]["reads_total"] == single["config"]["reads_total"] == single["config"]["reads_per_gpu"]
    assert abs(multi["config"]["reads_per_gpu"] * world - multi["config"]["reads_total"]) <= world
    assert multi["sketch_check"]["n_hashes"] == 1000
    assert _fp(multi) == _fp(single)
    # value = total bases * steps / time, not per-rank bases
    assert abs(multi["value"] * multi["ms_per_step"] / 1e3 - multi["config"]["reads_total"] * 150) < 1e-3 * multi["config"]["reads_total"] * 150


def test_driver_shaped_launch_of_two_gpus_matches_the_oracle_golden_at_full_size():
    """`python bench.py --gpus 2` exactly as the driver would type it (no launcher in front), BASELINE configs[3] at its full
    50 Gbase, the two read blocks resident on the one GPU of this box: the line must name configs[3] and carry
    matches_golden = true (the fingerprint the oracle computed on the CPU for the whole read set)."""
    out = _bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], 2, launcher=False)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert "BASELINE configs[3]" in out["config"]["workload"] and out["config"]["reads_total"] == 333333334
    assert out["sketch_check"]["matches_golden"] is True and out["sketch_check"]["golden"]


def test_eight_read_blocks_in_one_library_call_match_the_oracle_golden_at_full_size():
    """`python bench.py --gpus 8`: BASELINE configs[3] as its north_star states it -- 50 Gbase in eight read blocks, one
    fh_sketch_device_blocks call per step (eight library threads, eight handles; here all on this box's one GPU), host merge --
    and the merged sketch is the oracle's"""
    out = _bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], 8, launcher=False)
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and "8 contiguous read blocks" in out["config"]["workload"]
    assert "BASELINE configs[3]" in out["config"]["workload"] and out["config"]["reads_total"] == 333333334
    assert out["sketch_check"]["matches_golden"] is True


def test_eight_ranks_under_the_launcher():
    """torch.distributed.run with eight ranks (gloo gather of the partial sketches, merge on rank 0) on a cut-down read set"""
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--gbases", "0.4"]
    multi = _bench(common, 8, True)
    single = _bench(common, 1)
    assert multi["n_gpus"] == 8 and _fp(multi) == _fp(single)


def test_single_gpu_default_is_configs3_and_checks_itself():
    out = _bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], 1)
    assert "BASELINE configs[3]" in out["config"]["workload"] and out["n_gpus"] == 1 and out["scaling"] == "strong"
    assert out["sketch_check"]["matches_golden"] is True
    # (reads of one length: the segment form of the kernel, fh_k2s.hip -- found by the block's own probe)
    assert out["roofline"]["frac"] > 0 and out["roofline"]["kernel"].startswith("k2_sketch_seg<21>")
    assert out["per_rank"] == [{"rank": 0, "reads": 333333334, "kernel_ms_per_pass": out["roofline"]["kernel_ms_per_pass"]}]


def test_c5_workload_two_handles_one_call():
    """--workload c5 on a cut-down batch (300 files: the 256 golden ones + a tail), devices = [0, 0] in ONE finch_sketch_files
    call: per-file sketches of the sample must give the oracle's fingerprint"""
    out = _bench(["--workload", "c5", "--files", "300", "--steps", "1", "--warmup", "0"], 2, launcher=False)
    assert out["config"]["files"] == 300 and out["n_gpus"] == 2
    assert out["sketch_check"]["sample_files"] == 256 and out["sketch_check"]["matches_golden"] is True
    assert out["value"] > 0 and out["config"]["files_per_s"] > 0
    # the batch's line carries the kernel's roofline block and the all-cores CPU baseline (one file per task)
    r, c = out["roofline"], out["cpu_baseline"]
    # (the files go many per launch -- fh_batch_*: far fewer sketch launches than files, every file accounted for)
    assert 1 <= r["launches"] < 300 and r["achieved"] > 0 and 0 < r["kernel_share_of_call"] <= 1.0
    assert r["files_taken_many_per_launch"] + r["files_through_own_sketcher"] == 300 and r["files_taken_many_per_launch"] >= 290
    assert r["pcie"]["achieved_gbs"] > 0
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "one file per task" in c["sample"]


def test_multi_rank_line_names_every_ranks_kernel_time():
    out = _bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--gbases", "0.8"], 4, launcher=False)
    assert [p["rank"] for p in out["per_rank"]] == [0, 1, 2, 3] and all(p["kernel_ms_per_pass"] > 0 for p in out["per_rank"])
    assert sum(p["reads"] for p in out["per_rank"]) == out["config"]["reads_total"]
    out = _bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--gbases", "0.8"], 2, launcher=True)
    assert [p["rank"] for p in out["per_rank"]] == [0, 1] and all(p["kernel_ms_per_pass"] > 0 for p in out["per_rank"])


def test_gather_on_rccl_with_device_tensors():
    """`--backend nccl` puts the gather of the partial sketches on RCCL with device tensors (sharding.gather_and_merge(device=
    "cuda")).  A box with one GPU cannot run two RCCL ranks; a world of one exercises the same code: group creation on the
    device, the tensor's trip to the GPU, dist.gather, the merge of what came back."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import finch_rs_amd as F
from finch_rs_amd import sketch_schemes as S, sharding as SH
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
p = F.SketchParams.mash(1000, 1000, True, 21, 0)
g = S.synth_genome_host(200000, 5)
sk = p.create_sketcher()
sk.push_block(S.synth_reads_host(g, 0, 5000, 150, 5, 10000, 500))
kc, km, pos = sk.to_arrays(); tk = sk.finish()[1]
mkc, mkm, mpos, mtk = SH.gather_and_merge(dist, p, (kc, km, pos, tk), 1000, device="cuda")
assert np.array_equal(mkc, kc) and np.array_equal(mkm, km) and np.array_equal(mpos, pos) and mtk == tk
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t.item()) == 1.5
dist.destroy_process_group()
print("rccl gather OK")
''' % (ROOT, str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "rccl gather OK" in r.stdout, r.stdout[-3000:]


def test_the_bench_line_collects_its_own_counters():
    """roofline.traffic of a standard-size run comes from PMC counters this very run collected (two profiled child runs), not
    from the committed passes; the counters must see about one byte per position (the stream is read once) and the VALU
    instruction count of the kernel that ran"""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("no rocprofv3 on this box")
    d = _bench(["--workload", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], 1)
    r = d["roofline"]
    assert r["traffic_source"].startswith("live"), r.get("traffic_source")
    live = r["pmc"]  # (the per-wave-iteration block of the line is this run's own; the committed passes' sit under pmc_committed)
    assert live["source"].startswith("live")
    assert 0.95 < live["hbm_bytes_per_position"] < 1.6, live
    assert 30.0 < live["valu_per_wave_iter"] < 80.0, live
    assert live["lds_active_per_wave_iter"] is None or live["lds_active_per_wave_iter"] > 0
    assert r["traffic"] == int(live["hbm_bytes_per_position"] * r["alg_bytes_per_launch"])
    committed = _bench(["--workload", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-live-pmc"], 1)
    assert committed["roofline"]["traffic_source"].startswith("committed")
