#!/bin/bash
# round 5, session f: quarter-wave LayerNorm + packed-fp32 GELU epilogue: kernel tests, A/B against the one-wave-per-row LN (ADA_LN_LPR=64) and the scalar GELU build
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5f
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r5f/tests.log 2>&1
tail -4 gpurun_out/r5f/tests.log
run() { "$@" timeout 600 python bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-low-mean 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['ms_per_step_repeats'], l['rel_l1'], l['roofline']['frac'])"; }
for i in 1 2; do
  echo "== default (quarter-wave LN for dim <= 512, packed GELU)"; run env
  echo "== ADA_LN_LPR=64"; run env ADA_LN_LPR=64
  echo "== scalar GELU build"; run env ADA_HIP_LIB=$PWD/amodal-depth-anything_amd/csrc/libada_hip_gelu0.so
done > gpurun_out/r5f/bench_ab.txt 2>&1
cat gpurun_out/r5f/bench_ab.txt
echo "== bench_gemm default"; REPS=30 timeout 300 python tools/bench_gemm.py 2>&1 | grep fc1
echo "== bench_gemm scalar GELU"; ADA_HIP_LIB=$PWD/amodal-depth-anything_amd/csrc/libada_hip_gelu0.so REPS=30 timeout 300 python tools/bench_gemm.py 2>&1 | grep fc1
