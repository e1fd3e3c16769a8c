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
    g = 0.06
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    multi = _bench(["--gbases", str(g)] + common, world)
    single = _bench(["--gbases", str(g * world)] + common, 1)
    assert multi["n_gpus"] == world and multi["steps"] == 2 and multi["scaling"] == "weak"
    assert multi["unit"] == "bases/s" and multi["value"] > 0 and multi["roofline"]["bound"] == "hbm"
    assert multi["sketch_check"]["n_hashes"] == 1000
    assert multi["sketch_check"] == single["sketch_check"]
