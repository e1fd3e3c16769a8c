#!/bin/bash
# round 4, sixth GPU call: static first units, flat flatten, register bitonic -- tests, the scaling proxy again, the batch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r04f
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for g in 50 6.25; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --gbases $g > $O/share_$g.json 2> $O/share_$g.err
  python -c "
import json
d = json.load(open('$O/share_$g.json'))
print('share $g Gbase: %.3f ms/step  %.1f Gbases/s  kernel %.3f ms/pass  golden %s' % (d['ms_per_step'], d['value'] / 1e9, d['roofline']['kernel_ms_per_pass'], d['sketch_check']['matches_golden']))"
done
FH_NO_STATIC_UNITS=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --gbases 6.25 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('no static units, 6.25 Gbase: %.3f ms/step kernel %.3f ms/pass' % (d['ms_per_step'], d['roofline']['kernel_ms_per_pass']))"
for t in 12 16; do python tools/batch_trace.py 1024 $t; done | tee $O/c5_threads.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/c5 -o t --output-format csv -- python $R/tools/batch_trace.py 1024 12 > $O/c5_trace.txt 2> $O/c5_trace.err
python $R/tools/trace_busy.py $O/c5 --tail 0.45 > $O/c5_busy.txt; cat $O/c5_trace.txt; head -16 $O/c5_busy.txt
rm -rf $O/c5
rocprofv3 --kernel-trace --stats -d $O/c4 -o t --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline --gbases 6.25 > $O/c4_6g.json 2> $O/c4_6g.err
python $R/tools/trace_busy.py $O/c4 --tail 0.5 --chain 14 > $O/c4_6g_timeline.txt; tail -15 $O/c4_6g_timeline.txt
rm -rf $O/c4
