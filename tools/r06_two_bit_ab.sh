#!/bin/bash
# configs[4] on one box, one finch_sketch_files call over 1 024 genomes (tools/batch_trace.py), three ways in turn, three rounds:
#   bytes on the link (option batch_two_bit=0: round 6's first form of the batch path)
#   the two-bit form written in two passes (strip, then pack: what a CPU without BMI2 runs; pack_scalar=2)
#   the two-bit form written in one pass (the default)
# then where a worker's time goes (option trace), and the 10 000-file call.
#   gpurun -- 'bash tools/r06_two_bit_ab.sh <tag>'   -> gpurun_out/<tag>_two_bit_ab.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r06}_two_bit_ab.txt; mkdir -p gpurun_out; : > $O
for i in 1 2 3; do
  echo "bytes on the link:        $(FH_DEBUG=batch_two_bit=0 python tools/batch_trace.py 1024 | tail -1)" >> $O
  echo "two-bit, two passes:      $(FH_DEBUG=pack_scalar=2 python tools/batch_trace.py 1024 | tail -1)" >> $O
  echo "two-bit, one pass:        $(python tools/batch_trace.py 1024 | tail -1)" >> $O
done
echo "--- a worker's time (option trace), bytes on the link / two-bit one pass" >> $O
FH_DEBUG=trace,batch_two_bit=0 python tools/batch_trace.py 1024 2>&1 | grep "worker 0" | tail -1 >> $O
FH_DEBUG=trace python tools/batch_trace.py 1024 2>&1 | grep "worker 0" | tail -1 >> $O
echo "--- 10 000 files" >> $O
echo "bytes on the link:        $(FH_DEBUG=batch_two_bit=0 python tools/batch_trace.py 10000 | tail -1)" >> $O
echo "two-bit, one pass:        $(python tools/batch_trace.py 10000 | tail -1)" >> $O
cat $O
# the kernel timeline of the timed 1 024-file call with the two-bit form (tools/trace_busy.py)   -> gpurun_out/<tag>_c5_busy_two_bit.txt
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/bt && rocprofv3 --kernel-trace -d /tmp/bt -o bt --output-format csv -- python $GRAFT_REPO_ROOT/tools/batch_trace.py 1024 > /tmp/bt_run.txt 2>&1 )
python tools/trace_busy.py /tmp/bt --tail 0.5 --chain 30 > gpurun_out/${1:-r06}_c5_busy_two_bit.txt 2>&1
tail -3 /tmp/bt_run.txt | cut -c1-200; head -30 gpurun_out/${1:-r06}_c5_busy_two_bit.txt
