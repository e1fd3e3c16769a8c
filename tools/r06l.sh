cd $GRAFT_REPO_ROOT
for pp in 33554432 8388608 4194304 16777216; do
  echo "== spec_prefix_pos=$pp"
  for g in 6.25 12.5 50; do
    FH_DEBUG=spec_prefix_pos=$pp python bench.py --gbases $g --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-live-pmc | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('  share %5s Gbase: %8.3f ms per pass  kernel %.3f ms  golden %s  launches %s' % ('$g', d['ms_per_step'], d['roofline']['kernel_ms_per_pass'], d['sketch_check']['matches_golden'], d['roofline']['launches']))"
  done
done 2>&1 | tee gpurun_out/r06l_ab_prefix.txt
