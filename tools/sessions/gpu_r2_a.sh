#!/bin/bash
# round-2 first GPU session: new kernels' parity tests, then A/B timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" > $O/test_attention.log 2>&1; echo "attention tests rc=$?" | tee -a $O/summary.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "forced_tile" > $O/test_tiles.log 2>&1; echo "tile tests rc=$?" | tee -a $O/summary.txt
for v in 3 0 1 3 0 1; do VARIANT=$v REPS=50 timeout 120 python tools/bench_attn.py 2>&1 | tail -1 | tee -a $O/attn_ab.txt; done
for v in 4 8 4 8; do ADA_IGEMM_VARIANT=$v REPS=20 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a $O/gemm_ab.txt; done
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_metrics.py -m gpu -x -q > $O/test_kernels_all.log 2>&1; echo "all kernel tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/test_attention.log $O/test_tiles.log $O/test_kernels_all.log
