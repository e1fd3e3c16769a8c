#!/bin/bash
# One parameterised GPU-box session script (replaces the one-off tools/gpu_r04*.sh lease scripts of round 4).
#   gpurun --timeout S -- 'bash tools/gpu_evidence.sh <tag> <stage> [<stage> ...]'      -> gpurun_out/<tag>_*
# stages (run in the order given):
#   box        what the box is (GPU, hardware threads, cgroup quota, tmpfs)
#   suite      the whole `-m gpu` suite with -rs (FH_REQUIRE_FULL=1: the *_full configs must run, not skip) + smoke()
#   bench      python bench.py --steps 20 --warmup 5 as the driver runs it (with extras)
#   bench8     --gpus 8 --share-gpu (one fh_sketch_device_blocks call per step, eight handles on the one GPU)
#   c5         python bench.py --workload c5 (10 000 files)
#   proxy      each rank's share of configs[3] alone on the GPU, predicted 1/2/4/8 efficiency, N handles on the one GPU
#   pmc:<name>:<key>[:bench args separated by ','] rocprofv3 stats + PMC sets of one bench command (tools/gpu_bench_full.sh)
#              e.g. pmc:c4:c4_k21_n1000   pmc:c3:c3_k31_n2000000:--workload,c2,--k,31,--n,2000000
#   ab:<ks>:<name=lib.so,...>[:n[:env]]   tools/ab_k.py over the named builds (default first = shipped library)
#   fuzz:<cases>:<seed>   the kernel fuzzer on fresh seeds
#   cmd:<shell command with , for spaces>  anything else, output to <tag>_cmd<i>.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
T=$1; shift
O=gpurun_out
mkdir -p $O
i=0
for st in "$@"; do
  i=$((i+1))
  IFS=':' read -r kind a b c d <<< "$st"
  case $kind in
    box)
      ( rocm-smi --showproductname 2>/dev/null | grep -i "card\|gfx" | head -4; echo "hardware threads: $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; df -h /dev/shm | tail -1; ls /sys/class/drm/ | tr '\n' ' ' ) > $O/${T}_box.txt 2>&1 ;;
    suite)
      FH_REQUIRE_FULL=1 timeout 2700 python -m pytest tests -x -q -m gpu -rs --durations=10 2>&1 | tail -30 | tee $O/${T}_pytest_gpu_tail.txt
      timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/${T}_smoke.txt ;;
    bench)
      timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err; echo "bench rc=$?"
      python tools/bench_brief.py $O/${T}_bench_default.json | tee $O/${T}_bench_brief.txt ;;
    bench8)
      timeout 600 python bench.py --gpus 8 --share-gpu --steps 5 --warmup 1 --no-extras --no-cpu-baseline > $O/${T}_bench_gpus8_share.json 2> $O/${T}_bench_gpus8_share.err; echo "bench8 rc=$?"
      python tools/bench_brief.py $O/${T}_bench_gpus8_share.json ;;
    c5)
      timeout 1500 python bench.py --workload c5 --steps 2 --warmup 1 > $O/${T}_bench_c5.json 2> $O/${T}_bench_c5.err; echo "c5 rc=$?"; tail -c 1500 $O/${T}_bench_c5.json ;;
    pmc)
      bash tools/gpu_bench_full.sh ${T}_$a $b ${c//,/ } > $O/${T}_${a}_full.log 2>&1; tail -4 $O/${T}_${a}_full.log
      rm -rf $O/${T}_${a}_stats $O/${T}_${a}_pmc_fetch $O/${T}_${a}_pmc_write $O/${T}_${a}_pmc_sq ;;
    proxy)
      python tools/scaling_proxy.py ${a:-20} 2>&1 | tee $O/${T}_scaling_proxy.txt ;;
    ab)
      libs="default="; [ -n "$b" ] && libs="$b"
      python tools/ab_k.py --libs "$libs" --ks "$a" --n ${c:-1000} --env "${d//,/ }" 2>&1 | tee $O/${T}_ab${i}.txt ;;
    fuzz)
      FH_FUZZ_CASES=${a:-500} FH_FUZZ_SEED=${b:-505050} timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4 | tee $O/${T}_fuzz.txt ;;
    cmd)
      bash -c "${a//,/ }" > $O/${T}_cmd${i}.txt 2>&1; tail -30 $O/${T}_cmd${i}.txt ;;
    *) echo "unknown stage $st" ;;
  esac
done
