cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_segments.py -x -q 2>&1 | tail -6
( python tools/ab_reads.py --ragged 35,150 --gbases 1.5 --ks 31,27,25 --steps 4
  python tools/ab_reads.py --ragged 35,150 --frac-full 0.8 --gbases 1.5 --ks 31 --steps 4
  python tools/ab_reads.py --ragged 100,250 --gbases 1.5 --ks 31 --steps 4
  python tools/ab_reads.py --len 150 --gbases 6 --ks 21,31 --steps 4 ) 2>&1 | tee gpurun_out/r06n_ab_ragged.txt
