#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2e
O=gpurun_out/r2e
for lib in libada_hip.so libada_hip_rs0.so libada_hip_rs2.so; do
  export ADA_HIP_LIB=$PWD/amodal-depth-anything_amd/csrc/$lib
  echo "== $lib" | tee -a $O/attn_rowsum.txt
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" 2>&1 | tail -n 2 | tee -a $O/attn_rowsum.txt
  for v in 3 5 0 3 5 0; do VARIANT=$v REPS=30 timeout 120 python tools/bench_attn.py 2>&1 | tail -n 1 | tee -a $O/attn_rowsum.txt; done
done
