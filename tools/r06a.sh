set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
python tools/h2d_rate.py > gpurun_out/r06a/h2d_rate.txt 2>&1
cat gpurun_out/r06a/h2d_rate.txt
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/bt -o bt --output-format csv -- python $GRAFT_REPO_ROOT/tools/batch_trace.py 1024 > $GRAFT_REPO_ROOT/gpurun_out/r06a/batch_trace.txt 2>&1 )
tail -3 gpurun_out/r06a/batch_trace.txt
python tools/trace_busy.py /tmp/bt --tail 0.5 --chain 30 > gpurun_out/r06a/c5_busy.txt 2>&1
cat gpurun_out/r06a/c5_busy.txt
python tools/batch_trace.py 1024 > gpurun_out/r06a/batch_plain.txt 2>&1; cat gpurun_out/r06a/batch_plain.txt
