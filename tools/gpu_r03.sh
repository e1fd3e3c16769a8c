#!/bin/bash
# round-3 GPU sessions, one stage per invocation:  bash tools/gpu_r03.sh <stage> [args]   (outputs -> gpurun_out/r03_<stage>*)
# (the stages are listed in profiles/README.md next to the files they produced)
STAGE=$1; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
case $STAGE in
ab_rounds)   # parity of the round structure + A/B of the variants
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "all_k or palindrom or golden or longer_sequence" 2>&1 | tail -5 | tee gpurun_out/r03_ab_rounds_pytest.txt
  timeout 900 python tools/ab_k.py --libs base=build/ab/base.so,r8=finch_rs_amd/libfinch_hip.so,r16=build/ab/r16.so,r8w5=build/ab/r8w5.so,r8from22=build/ab/r8from22.so \
      --ks 21,22,24,25,28,31,32 2>&1 | tee gpurun_out/r03_ab_rounds.txt
  ;;
suite)       # box facts, the whole GPU suite, the default bench line, the driver-shaped 2-"GPU" launch
  (nproc; free -g | head -2; df -h /dev/shm /tmp | cat; rocm-smi --showmeminfo vram 2>/dev/null | head -8) > gpurun_out/r03_box.txt 2>&1
  timeout 2400 python -m pytest tests -x -q -m gpu -rs --durations=15 2>&1 | tail -40 | tee gpurun_out/r03_pytest_gpu_tail.txt
  timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
  timeout 600 python bench.py --gpus 2 --share-gpu --steps 5 --warmup 1 > gpurun_out/r03_bench_gpus2_share.json 2> gpurun_out/r03_bench_gpus2_share.err; echo "bench2 rc=$?"
  ;;
pmc)         # rocprofv3 kernel stats + PMC passes: configs[3] (headline), k=31 (spill-free now), and the two-word kernels
  bash tools/gpu_bench_full.sh r03_c4 c4_k21_n1000
  bash tools/gpu_bench_full.sh r03_k31 c2_k31_n1000 --workload c2 --k 31
  bash tools/gpu_bench_full.sh r03_k33 c2_k33_n1000 --workload c2 --k 33
  bash tools/gpu_bench_full.sh r03_k64 c2_k64_n1000 --workload c2 --k 64
  ;;
ab_lds)      # wide LDS reads (16-byte A records, 8-byte B records everywhere) vs round 2's build and vs 4-byte B reads; fwd-only ceiling
  timeout 900 python tools/ab_k.py --libs r02=build/ab/base.so,cur=finch_rs_amd/libfinch_hip.so,nob64=build/ab/nob64.so,fwdonly=build/ab/fwdonly.so \
      --ks 16,21,24,27,31,32,33,48,64 2>&1 | tee gpurun_out/r03_ab_lds.txt
  ;;
e2e)         # where the end-to-end paths spend their time
  python tools/e2e_trace.py 4000000 1,4,8,16 > gpurun_out/r03_e2e_trace.txt 2> gpurun_out/r03_e2e_trace.err
  python tools/one_worker_trace.py > gpurun_out/r03_one_worker.txt 2>&1
  python tools/batch_threads.py > gpurun_out/r03_batch_threads.txt 2>&1
  ;;
c3prof)      # configs[2] (k=31, 2 M hashes): phases, kernel timeline, counters of the sketch launches
  FH_DEBUG=trace python tools/phase_times.py --k 31 --n 2000000 --reps 3 > gpurun_out/r03_c3_phases.txt 2>&1
  cd /tmp; rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r03_c3_trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/phase_times.py --k 31 --n 2000000 --reps 2 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  python tools/kernel_timeline.py gpurun_out/r03_c3_trace --min-ms 0.05 > gpurun_out/r03_c3_kernel_timeline.txt 2>&1
  bash tools/pmc_k2.sh r03_c3 python $GRAFT_REPO_ROOT/tools/phase_times.py --k 31 --n 2000000 --reps 1 > gpurun_out/r03_c3_pmc.txt 2>&1
  ;;
round2)      # after: interpolated sample threshold, H2D prefetch, wide LDS reads -- parity first, then the numbers
  timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_layer.py tests/test_gpu_fuzz.py tests/test_gpu_errors.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r03_round2_pytest.txt
  python tools/e2e_trace.py 4000000 1,8 > gpurun_out/r03b_e2e_trace.txt 2> gpurun_out/r03b_e2e_trace.err
  FH_DEBUG=trace python tools/phase_times.py --k 31 --n 2000000 --reps 3 2>&1 | grep -v "launch " > gpurun_out/r03b_c3_phases.txt
  FH_DEBUG=sample_want=1.25,trace python tools/phase_times.py --k 31 --n 2000000 --reps 3 2>&1 | grep -v "launch " > gpurun_out/r03b_c3_phases_want125.txt
  bash tools/pmc_k2.sh r03b_c3 python $GRAFT_REPO_ROOT/tools/phase_times.py --k 31 --n 2000000 --reps 1 > gpurun_out/r03b_c3_pmc.txt 2>&1
  ;;
c5prof)      # what the GPU does per file of a batch (configs[4]'s shape)
  for nt in 4 8 12 16 24; do python tools/batch_trace.py 512 $nt; done > gpurun_out/r03_c5_threads.txt 2>&1
  cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_c5_trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/batch_trace.py 512 12 > $GRAFT_REPO_ROOT/gpurun_out/r03_c5_trace.log 2>&1; cd $GRAFT_REPO_ROOT
  cp gpurun_out/r03_c5_trace/t_kernel_stats.csv gpurun_out/r03_c5_kernel_stats.csv
  python - <<'PY' > gpurun_out/r03_c5_busy.txt
import csv
rows = sorted(csv.DictReader(open("gpurun_out/r03_c5_trace/t_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# the timed call = the last 60 % of the trace's kernels; union of busy intervals over it
ev = ev[len(ev) * 2 // 5:]
t0, t1 = ev[0][0], max(e for _, e in ev)
busy, cur_s, cur_e = 0, None, None
for s, e in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e in ev)
print("kernels %d  span %.1f ms  union-busy %.1f ms (%.0f %%)  sum of durations %.1f ms (overlap factor %.2f)" % (len(ev), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), tot / 1e6, tot / max(busy, 1)))
PY
  ;;
round3)      # after: inline single-chunk pump, one-block copy-out, two-stage sample cap
  timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_layer.py tests/test_gpu_fuzz.py tests/test_gpu_errors.py tests/test_gpu_bgzf_device.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r03_round3_pytest.txt
  for nt in 8 12 16; do python tools/batch_trace.py 512 $nt; done > gpurun_out/r03c_c5_threads.txt 2>&1
  python tools/one_worker_trace.py 2>&1 | tail -4 > gpurun_out/r03c_one_worker.txt
  FH_DEBUG=trace python tools/phase_times.py --k 31 --n 2000000 --reps 3 2>&1 | grep -v "launch " > gpurun_out/r03c_c3_phases.txt
  ;;
c5bench)     # configs[4] at its size through bench.py (one GPU; and two handles on the one GPU), multirank tests
  timeout 1500 python bench.py --workload c5 --steps 2 --warmup 1 > gpurun_out/r03_bench_c5.json 2> gpurun_out/r03_bench_c5.err; echo "c5 rc=$?"
  timeout 900 python bench.py --workload c5 --files 2000 --gpus 2 --share-gpu --steps 2 --warmup 1 > gpurun_out/r03_bench_c5_gpus2_share.json 2> gpurun_out/r03_bench_c5_gpus2_share.err; echo "c5x2 rc=$?"
  timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r03_multirank_pytest.txt
  ;;
round4)      # after: lazy wide-column copy-out (+ device-side row gather)
  timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_layer.py tests/test_gpu_full_size.py -x -q -m gpu -k "not c4 and not c5 and not c2" 2>&1 | tail -4 | tee gpurun_out/r03_round4_pytest.txt
  python tools/c3_resident.py > gpurun_out/r03d_c3_resident.txt 2>&1
  FH_DEBUG=no_lazy_copyout python tools/c3_resident.py > gpurun_out/r03d_c3_resident_nolazy.txt 2>&1
  ;;
final)       # the state the round ends in
  timeout 2400 python -m pytest tests -x -q -m gpu -rs --durations=8 2>&1 | tail -24 | tee gpurun_out/r03z_pytest_gpu_tail.txt
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/r03z_smoke.txt
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03z_bench_default.json 2> gpurun_out/r03z_bench_default.err; echo "bench rc=$?"
  timeout 600 python bench.py --gpus 2 --share-gpu --steps 5 --warmup 1 > gpurun_out/r03z_bench_gpus2_share.json 2> gpurun_out/r03z_bench_gpus2_share.err; echo "bench2 rc=$?"
  cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03z_stats -o stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r03z_stats.log 2>&1; cd $GRAFT_REPO_ROOT
  cp gpurun_out/r03z_stats/stats_kernel_stats.csv gpurun_out/r03z_kernel_stats.csv
  ;;
ab_waves)    # k <= 24 in rounds of 16 / 8 at four and five waves per SIMD (LDS 32 KB per workgroup makes five fit)
  L=cur=finch_rs_amd/libfinch_hip.so,r16=build/ab/k21r16.so,r16w5=build/ab/k21r16w5.so,r8w5=build/ab/k21r8w5.so
  timeout 900 python tools/ab_k.py --libs $L --ks 17,20,21,24,31 2>&1 | tee gpurun_out/r03_ab_waves.txt
  timeout 900 python tools/ab_k.py --libs $L --ks 17,20,21,24,31 --env "FH_DEBUG=waves_per_cu=20" 2>&1 | tee gpurun_out/r03_ab_waves_20percu.txt
  ;;
round5)      # after: fused filter passes (finish_mash_in_place)
  timeout 1800 python -m pytest tests/test_gpu_host_layer.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -x -q -m gpu -k "not c4 and not c5 and not c2" 2>&1 | tail -4 | tee gpurun_out/r03_round5_pytest.txt
  python tools/c3_resident.py > gpurun_out/r03e_c3_resident.txt 2>&1
  FUZZ_CASES=150 timeout 900 python tools/fuzz_params.py > gpurun_out/r03e_fuzz_params.txt 2>&1; tail -3 gpurun_out/r03e_fuzz_params.txt
  ;;
fuzz)        # a fuzz campaign over the paths round 3 touched (inline pump, prefetch, sharded reader, copy-out forms, kernels)
  ( timeout 2400 python tools/fuzz_params.py 1200 930001 2>&1 | tail -2
    FUZZ_FILES=1 timeout 2400 python tools/fuzz_device_text.py 1200 930002 2>&1 | tail -2
    FUZZ_SHARDED=1 timeout 2400 python tools/fuzz_device_text.py 1200 930003 2>&1 | tail -2
    FH_FUZZ_CASES=2000 FH_FUZZ_SEED=31337 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2 ) | tee gpurun_out/r03_fuzz_campaign.txt
  ;;
*) echo "unknown stage $STAGE"; exit 2;;
esac
