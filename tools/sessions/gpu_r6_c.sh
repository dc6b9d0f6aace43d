#!/bin/bash
# round 6, session c: fixed kernel tests of the split outputs, the third rung's accuracy on raw ViT-G (HEAD=split), head streams A/B (config 2, B = 1),
# configs 2 / 5 (centred + un-centred twin), the packed-fp16 exp2 micro-benchmark
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r6c
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_f8.py tests/test_gpu_model.py -m gpu -q -s -p no:cacheprovider -k "split_output or saturation or f8_terms or class_tokens or eight_ranks" 2>&1 | grep -v amdgpu | grep -v "^$" | tail -25
KS=8,40 HEAD=split timeout 900 python tools/stage_errors.py raw_vitg_224 raw_vitg_126x154_unc raw_vitg_126x154_unc_w1 2>&1 | grep -v amdgpu > gpurun_out/r6c/stage_errors_vitg_head_split.txt
grep "^#\|stage\|tap3\|out" gpurun_out/r6c/stage_errors_vitg_head_split.txt
for hs in 0 1; do echo "ADA_HEAD_STREAMS=$hs"; ADA_HEAD_STREAMS=$hs ENCODER=vitb B=8 timeout 600 python tools/config_shapes.py 2>&1 | grep "whole forward"; ADA_HEAD_STREAMS=$hs timeout 600 python tools/latency_b1.py vitb vitl 2>&1 | grep "B="; done > gpurun_out/r6c/head_streams_ab.txt 2>&1
cat gpurun_out/r6c/head_streams_ab.txt
timeout 1200 python tools/run_configs.py 2>&1 | grep -v amdgpu > gpurun_out/r6c/other_configs.txt; cat gpurun_out/r6c/other_configs.txt
timeout 300 tools/ubench/softmax_slot > gpurun_out/r6c/softmax_slot_packed_exp2.txt 2>&1; cat gpurun_out/r6c/softmax_slot_packed_exp2.txt
