#!/bin/bash
# round 5, session c: persistent 4-wave GEMM with the conflict-free slab layout: tests, isolated shapes, whole step A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5c
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "forced_tile or persistent" -p no:cacheprovider > gpurun_out/r5c/tests.log 2>&1
tail -5 gpurun_out/r5c/tests.log
for v in 0 16 32; do
  echo "== bench_gemm ADA_IGEMM_VARIANT=$v" ; ADA_IGEMM_VARIANT=$v REPS=30 timeout 300 python tools/bench_gemm.py 2>&1 | tail -5
done > gpurun_out/r5c/bench_gemm.txt 2>&1
cat gpurun_out/r5c/bench_gemm.txt
for v in 0 16 32; do
  echo "== bench_vs_lib ADA_IGEMM_VARIANT=$v"; ADA_IGEMM_VARIANT=$v REPS=30 timeout 300 python tools/bench_vs_lib.py 2>&1 | grep gemm
done > gpurun_out/r5c/vs_lib.txt 2>&1
cat gpurun_out/r5c/vs_lib.txt
for v in 0 32 0 32; do
  echo "== bench.py ADA_IGEMM_VARIANT=$v"; ADA_IGEMM_VARIANT=$v timeout 600 python bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-low-mean 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['ms_per_step_repeats'], l['rel_l1'], l['roofline']['frac'], l['roofline']['avg_launch_ms'])"
done > gpurun_out/r5c/bench_ab.txt 2>&1
cat gpurun_out/r5c/bench_ab.txt
