set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -15
python tools/batch_trace.py 1024 > gpurun_out/r06b/batch_plain.txt 2>&1; cat gpurun_out/r06b/batch_plain.txt
python tools/batch_trace.py 1024 > gpurun_out/r06b/batch_plain2.txt 2>&1; cat gpurun_out/r06b/batch_plain2.txt
FH_DEBUG=file_batch=0 python tools/batch_trace.py 1024 > gpurun_out/r06b/batch_plain_off.txt 2>&1; cat gpurun_out/r06b/batch_plain_off.txt
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/bt -o bt --output-format csv -- python $GRAFT_REPO_ROOT/tools/batch_trace.py 1024 > $GRAFT_REPO_ROOT/gpurun_out/r06b/batch_trace.txt 2>&1 )
tail -3 gpurun_out/r06b/batch_trace.txt
python tools/trace_busy.py /tmp/bt --tail 0.5 --chain 30 > gpurun_out/r06b/c5_busy.txt 2>&1
cat gpurun_out/r06b/c5_busy.txt
python -m pytest tests/test_gpu_host_layer.py -x -q 2>&1 | tail -5
