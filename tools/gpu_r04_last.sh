#!/bin/bash
# round 4, the state the round ends in (tag r04zz): the whole GPU suite, smoke, the driver-shaped bench line, eight handles on
# the one GPU.  (The kernels of K <= 22 and K >= 29 -- configs[3], configs[1], k = 31, configs[2] -- are the ones r04z's
# rocprofv3 / PMC sets were taken from; K = 23..28 changed since: profiles/r04j_ab_pre_wide.txt.)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
T=r04zz
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -rs --durations=10 2>&1 | tail -26 | tee $O/${T}_pytest_gpu_tail.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/${T}_smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 8 --share-gpu --steps 5 --warmup 1 --no-extras --no-cpu-baseline > $O/${T}_bench_gpus8_share.json 2> $O/${T}_bench_gpus8_share.err; echo "bench8 rc=$?"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/${T}_stats -o stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/$O/${T}_stats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $O/${T}_stats/stats_kernel_stats.csv $O/${T}_c4_kernel_stats.csv 2>/dev/null; rm -rf $O/${T}_stats
python - <<PY
import json
d = json.load(open("$O/${T}_bench_default.json"))
print("default: %.1f Gbases/s %.3f ms golden %s cpu %s" % (d["value"] / 1e9, d["ms_per_step"], d["sketch_check"]["matches_golden"], d["cpu_baseline"]["value"]))
for k, v in d.get("extras", {}).items():
    print("  ", k, {kk: vv for kk, vv in v.items() if kk not in ("what", "pmc", "sketch_check")}, (v.get("sketch_check") or {}).get("matches_golden"))
d = json.load(open("$O/${T}_bench_gpus8_share.json"))
print("gpus8 share: %.1f Gbases/s golden %s" % (d["value"] / 1e9, d["sketch_check"]["matches_golden"]))
PY
