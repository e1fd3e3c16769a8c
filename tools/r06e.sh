cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( python tools/ab_reads.py --len 250 --gbases 6 --ks 21,31 --steps 4
  python tools/ab_reads.py --len 300 --gbases 6 --ks 21,31 --steps 4
  python tools/ab_reads.py --len 250 --gbases 6 --ks 21 --steps 4 --seed 42
  python tools/ab_reads.py --len 150 --gbases 6 --ks 21,31 --steps 4 --seed 42
  python tools/ab_reads.py --len 150 --gbases 6 --ks 21,31 --steps 4 ) 2>&1 | tee gpurun_out/r06e_ab_long_reads.txt
