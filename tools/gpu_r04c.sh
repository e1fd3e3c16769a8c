#!/bin/bash
# round 4, third GPU call: the scaling proxy (each rank's share of configs[3] alone on the one GPU), N handles sharing the
# GPU through the library call, the default bench line with extras, and a kernel trace of it
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
nproc > $O/box.txt
for g in 50 25 12.5 6.25; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --gbases $g > $O/share_$g.json 2> $O/share_$g.err
  python - <<PY
import json
d = json.load(open("$O/share_$g.json"))
print("share $g Gbase: %.3f ms/step  %.1f Gbases/s  kernel %.3f ms/pass  launches/pass %.2f" % (d["ms_per_step"], d["value"] / 1e9, d["roofline"]["kernel_ms_per_pass"], d["roofline"]["launches"] / d["steps"]))
PY
done
for n in 2 4 8; do
  timeout 900 python bench.py --gpus $n --share-gpu --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/threads_$n.json 2> $O/threads_$n.err
  python -c "import json; d=json.load(open('$O/threads_$n.json')); print('threads N=$n share-gpu: %.3f ms/step  %.1f Gbases/s  golden %s' % (d['ms_per_step'], d['value']/1e9, d['sketch_check']['matches_golden']))"
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --share-gpu --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/torchrun_4.json 2> $O/torchrun_4.err
grep '^{' $O/torchrun_4.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('torchrun N=4 share-gpu: %.3f ms/step  %.1f Gbases/s  golden %s' % (d['ms_per_step'], d['value']/1e9, d['sketch_check']['matches_golden']))"
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("default: %.1f Gbases/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
for k, v in d.get("extras", {}).items():
    print("  ", k, {kk: vv for kk, vv in v.items() if kk not in ("what", "pmc", "sketch_check")}, (v.get("sketch_check") or {}).get("matches_golden"))
PY
cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o c4 -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline > $OLDPWD/$O/prof_bench.json 2> $OLDPWD/$O/prof_bench.err; cd $OLDPWD
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -12 $O/kernel_stats.csv
