#!/bin/bash
# round 4 evidence: whole GPU suite, smoke, the driver-shaped bench lines, rocprofv3 stats + PMC sets of configs[3] and k = 31
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
T=${1:-r04a}
O=gpurun_out
mkdir -p $O
( rocm-smi --showproductname 2>/dev/null | grep -i "card\|gfx" | head -4; echo "hardware threads: $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; df -h /dev/shm | tail -1 ) > $O/${T}_box.txt 2>&1
timeout 2700 python -m pytest tests -x -q -m gpu -rs --durations=10 2>&1 | tail -26 | tee $O/${T}_pytest_gpu_tail.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/${T}_smoke.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --gpus 8 --share-gpu --steps 5 --warmup 1 --no-extras --no-cpu-baseline > $O/${T}_bench_gpus8_share.json 2> $O/${T}_bench_gpus8_share.err; echo "bench8 rc=$?"
bash tools/gpu_bench_full.sh ${T}_c4 c4_k21_n1000 > $O/${T}_c4_full.log 2>&1; tail -3 $O/${T}_c4_full.log
bash tools/gpu_bench_full.sh ${T}_k31 c2_k31_n1000 --workload c2 --k 31 > $O/${T}_k31_full.log 2>&1; tail -3 $O/${T}_k31_full.log
timeout 900 python bench.py --workload c5 --steps 2 --warmup 1 > $O/${T}_bench_c5.json 2> $O/${T}_bench_c5.err; echo "c5 rc=$?"; tail -c 600 $O/${T}_bench_c5.json
rm -rf $O/${T}_*_stats $O/${T}_*_pmc_fetch $O/${T}_*_pmc_write $O/${T}_*_pmc_sq
python - <<PY
import json
d = json.load(open("$O/${T}_bench_default.json"))
print("default: %.1f Gbases/s %.3f ms golden %s cpu %s" % (d["value"] / 1e9, d["ms_per_step"], d["sketch_check"]["matches_golden"], d["cpu_baseline"]["value"]))
for k, v in d.get("extras", {}).items():
    print("  ", k, {kk: vv for kk, vv in v.items() if kk not in ("what", "pmc", "sketch_check")}, (v.get("sketch_check") or {}).get("matches_golden"))
PY
