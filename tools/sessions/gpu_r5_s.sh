#!/bin/bash
# round 5, session s: the 4-wave loop for cheap-epilogue launches from K = 1536 (raw ViT-G qkv / w12) -- kernel tests, ViT-G fixtures, configs, headline unchanged?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5s
O=gpurun_out/r5s
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f8.py -q -x -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -n 4
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden and vitg" -p no:cacheprovider 2>&1 | grep "rel-L1\|passed\|failed" | sed 's/^\.//' > $O/vitg_parity.txt; cat $O/vitg_parity.txt
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu > $O/configs.txt; cat $O/configs.txt
ADA_IGEMM_VARIANT=4 timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu > $O/configs_8wave_only.txt; cat $O/configs_8wave_only.txt
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu > $O/configs_again.txt; cat $O/configs_again.txt
