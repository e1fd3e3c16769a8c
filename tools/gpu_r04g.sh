#!/bin/bash
# round 4: small FASTA packed on the host + reset folded into fh_finish's epilogue
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r04g
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_parity.py tests/test_gpu_host_layer.py tests/test_gpu_multirank.py -x -q -k "not full_size" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for t in 12 16 24; do python tools/batch_trace.py 1024 $t; done | tee $O/c5_threads.txt
for t in 12 16; do FINCH_SMALL_FASTA_HOST=0 python tools/batch_trace.py 1024 $t; done | tee $O/c5_threads_device_parse.txt
FH_NO_RESET_FOLD=1 python tools/batch_trace.py 1024 16 | tee $O/c5_no_fold.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/c5 -o t --output-format csv -- python $R/tools/batch_trace.py 1024 16 > $O/c5_trace.txt 2> $O/c5_trace.err
python $R/tools/trace_busy.py $O/c5 --tail 0.45 > $O/c5_busy.txt; cat $O/c5_trace.txt; head -12 $O/c5_busy.txt
rm -rf $O/c5
rocprofv3 --kernel-trace --stats -d $O/c5one -o t --output-format csv -- python $R/tools/batch_trace.py 96 1 > /dev/null 2>&1
python $R/tools/trace_busy.py $O/c5one --tail 0.3 --chain 12 | tail -14
rm -rf $O/c5one
cd $R
python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --gbases 6.25 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('share 6.25: %.3f ms/step kernel %.3f ms/pass' % (d['ms_per_step'], d['roofline']['kernel_ms_per_pass']))"
