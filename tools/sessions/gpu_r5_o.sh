#!/bin/bash
# round 5, session o: fp8 correction terms -- kernel tests, raw ViT-G parity with the terms on the fp8 pipe and on the fp16 pipe, config 5 both ways, headline unchanged?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5o
O=gpurun_out/r5o
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_f8.py -q -s -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -n 60 > $O/f8_tests.txt; tail -n 25 $O/f8_tests.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -n 6 > $O/kernel_tests.txt; tail -n 4 $O/kernel_tests.txt
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden and vitg" -p no:cacheprovider 2>&1 | grep "rel-L1\|passed\|failed" | sed 's/^\.//' > $O/vitg_parity_f8.txt; cat $O/vitg_parity_f8.txt
ADA_F8_CORR=0 timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden and vitg" -p no:cacheprovider 2>&1 | grep "rel-L1\|passed\|failed" | sed 's/^\.//' > $O/vitg_parity_fp16_terms.txt; cat $O/vitg_parity_fp16_terms.txt
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu > $O/configs_f8.txt; cat $O/configs_f8.txt
ADA_F8_CORR=0 timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu > $O/configs_fp16_terms.txt; cat $O/configs_fp16_terms.txt
timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-low-mean > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
