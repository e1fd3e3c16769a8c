cd $GRAFT_REPO_ROOT
( echo "== 8 size classes (default build)"; python tools/ab_reads.py --ragged 35,150 --gbases 1.5 --ks 31,27 --steps 5
  echo "== 4 size classes"; FH_LIB=$GRAFT_REPO_ROOT/finch_rs_amd/libfinch_hip_c4.so FH_NO_AUTOBUILD=1 python tools/ab_reads.py --ragged 35,150 --gbases 1.5 --ks 31,27 --steps 5
  echo "== 8 again"; python tools/ab_reads.py --ragged 35,150 --gbases 1.5 --ks 31 --steps 5
  python tools/ab_reads.py --ragged 35,150 --frac-full 0.8 --gbases 1.5 --ks 31 --steps 5 ) 2>&1 | tee gpurun_out/r06o_ab_rag_classes.txt
