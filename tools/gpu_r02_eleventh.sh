#!/bin/bash
# kernel timeline of the device-side BGZF path
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
for mode in "" "--noisy"; do
  tag=const; [ -n "$mode" ] && tag=noisy
  rm -rf /tmp/prof_bz
  rocprofv3 --kernel-trace -d /tmp/prof_bz -o bz --output-format csv -- python $GRAFT_REPO_ROOT/tools/bgzf_device_file.py $mode > /tmp/bz_$tag.log 2>&1
  (cd $GRAFT_REPO_ROOT && tail -3 /tmp/bz_$tag.log && python tools/kernel_timeline.py /tmp/prof_bz --min-ms 0.05 | tail -40) > $GRAFT_REPO_ROOT/gpurun_out/r02k_bgzf_timeline_$tag.txt 2>&1
  cat $GRAFT_REPO_ROOT/gpurun_out/r02k_bgzf_timeline_$tag.txt
done
