#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4q
O=$PWD/gpurun_out/r4q
echo "== reader: sc1 loads"; timeout 300 python tools/bench_ln_tail.py 2>&1 | grep -v amdgpu | tee $O/ln_tail_anatomy_sc1.txt
echo "== reader: acquire + plain loads"; ADA_HIP_LIB=$PWD/amodal-depth-anything_amd/csrc/libada_hip_lnplain.so timeout 300 python tools/bench_ln_tail.py 2>&1 | grep -v amdgpu | tee $O/ln_tail_anatomy_plain.txt
ADA_HIP_LIB=$PWD/amodal-depth-anything_amd/csrc/libada_hip_lnplain.so timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "layernorm_tail" 2>&1 | tail -2
