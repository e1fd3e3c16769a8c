#!/bin/bash
# round 4, fourth GPU call: where the time of a pass / of a file of a batch goes (kernel timelines), and the batch's thread count
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r04d
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/c4 -o t --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline --gbases 6.25 > $O/c4_6g.json 2> $O/c4_6g.err
python $R/tools/trace_busy.py $O/c4 --tail 0.5 --chain 24 > $O/c4_6g_timeline.txt; cat $O/c4_6g_timeline.txt
rocprofv3 --kernel-trace --stats -d $O/c5one -o t --output-format csv -- python $R/tools/batch_trace.py 96 1 > $O/c5_one_thread.txt 2> $O/c5one.err
python $R/tools/trace_busy.py $O/c5one --tail 0.6 --chain 40 > $O/c5_one_thread_timeline.txt; cat $O/c5_one_thread.txt; head -70 $O/c5_one_thread_timeline.txt
rocprofv3 --kernel-trace --stats -d $O/c5twelve -o t --output-format csv -- python $R/tools/batch_trace.py 1024 12 > $O/c5_12_threads.txt 2> $O/c5twelve.err
python $R/tools/trace_busy.py $O/c5twelve --tail 0.9 > $O/c5_12_threads_busy.txt; cat $O/c5_12_threads.txt; head -24 $O/c5_12_threads_busy.txt
cd $R
for t in 8 12 16 24 32; do python tools/batch_trace.py 1024 $t; done | tee $O/c5_threads.txt
rm -rf $O/c4 $O/c5one $O/c5twelve
