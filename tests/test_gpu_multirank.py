"""bench.py's N > 1 flow on a 1-GPU box: two and three ranks share cuda:0 (--share-gpu), each sketches its own read
block, rank 0 gathers and merges.  The merged sketch of N ranks x G Gbase must be the sketch one rank computes on
N*G Gbase (same read indices): SURVEY 8e through the real launcher, timing protocol and JSON contract included."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(args, world=1):
    if world > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", str(world), "--share-gpu"]
    else:
        cmd = [sys.executable, "bench.py"]
    r = subprocess.run(cmd + args, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_times_block_equals_one_rank_on_everything(world):
    """--workload c2 (weak scaling): every rank its own G Gbase"""
    g = 0.06
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    multi = _bench(["--workload", "c2", "--gbases", str(g)] + common, world)
    single = _bench(["--gbases", str(g * world)] + common, 1)
    assert multi["n_gpus"] == world and multi["steps"] == 2 and multi["scaling"] == "weak"
    assert multi["unit"] == "bases/s" and multi["value"] > 0 and multi["roofline"]["bound"] == "hbm"
    assert multi["sketch_check"]["n_hashes"] == 1000
    assert multi["sketch_check"] == single["sketch_check"]


@pytest.mark.parametrize("world", [2, 3])
def test_c4_workload_splits_one_read_set_into_read_blocks(world):
    """the default for N > 1 is BASELINE configs[3]'s shape (strong scaling): G Gbase IN TOTAL, rank r takes the read
    block shard_bounds(R, r, N); the merged sketch must be the one a single rank computes on all R reads, and `value`
    counts the total once."""
    g = 0.2
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--gbases", str(g)]
    multi = _bench(common, world)  # (--workload defaults to c4 when N > 1)
    single = _bench(["--workload", "c4"] + common, 1)
    assert multi["n_gpus"] == world and multi["scaling"] == "strong" and single["scaling"] == "strong"
    assert "configs[3] generator" in multi["config"]["workload"] and "%d contiguous read blocks" % world in multi["config"]["workload"]
    assert multi["config"]["reads_total"] == single["config"]["reads_total"] == single["config"]["reads_per_gpu"]
    assert abs(multi["config"]["reads_per_gpu"] * world - multi["config"]["reads_total"]) <= world
    assert multi["sketch_check"]["n_hashes"] == 1000
    assert multi["sketch_check"] == single["sketch_check"]
    # value = total bases * steps / time, not per-rank bases
    assert abs(multi["value"] * multi["ms_per_step"] / 1e3 - multi["config"]["reads_total"] * 150) < 1e-3 * multi["config"]["reads_total"] * 150
