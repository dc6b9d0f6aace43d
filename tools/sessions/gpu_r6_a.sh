#!/bin/bash
# round 6, session a: the pruned build (ABI 8) -- per-stage error of raw ViT-G against the oracle as the number of split blocks grows (the non-monotone
# parity of VERDICT r5 weak 2), the self-calibration table, then the GPU suite and a short bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r6a
export HSA_ENABLE_IPC_MODE_LEGACY=0
KS=0,8,12,16,40 F8=none timeout 900 python tools/stage_errors.py raw_vitg_224 > gpurun_out/r6a/stage_errors_vitg_fp16terms.txt 2>&1
tail -12 gpurun_out/r6a/stage_errors_vitg_fp16terms.txt
KS=0,8,12,16,40 timeout 900 python tools/stage_errors.py raw_vitg_224 raw_vitg_224_w1 > gpurun_out/r6a/stage_errors_vitg_policy.txt 2>&1
tail -8 gpurun_out/r6a/stage_errors_vitg_policy.txt
timeout 900 python tools/calibration_table.py > gpurun_out/r6a/calibration_table.txt 2>&1
cat gpurun_out/r6a/calibration_table.txt | tail -45
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=15 > gpurun_out/r6a/gpu_suite.txt 2>&1
tail -40 gpurun_out/r6a/gpu_suite.txt
timeout 600 python bench.py --no-cpu-baseline --no-traffic > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
tail -c 1500 gpurun_out/r6a/bench.json
