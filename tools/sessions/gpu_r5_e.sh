#!/bin/bash
# round 5, session e: per-launch tables of config 2 (ViT-B, 8 x 518^2) with a tile sweep, B = 1, the ViT-L bs=32 sweep, single-image latency, other configs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5e
export HSA_ENABLE_IPC_MODE_LEGACY=0
ENCODER=vitb B=8 SWEEP=1 timeout 900 python tools/config_shapes.py > gpurun_out/r5e/config2_shapes.txt 2>&1
head -4 gpurun_out/r5e/config2_shapes.txt
ENCODER=vitb B=1 SWEEP=1 timeout 900 python tools/config_shapes.py > gpurun_out/r5e/vitb_b1_shapes.txt 2>&1
head -4 gpurun_out/r5e/vitb_b1_shapes.txt
ENCODER=vitl B=32 SWEEP=1 REPS=5 timeout 1200 python tools/config_shapes.py > gpurun_out/r5e/vitl_b32_shapes.txt 2>&1
head -4 gpurun_out/r5e/vitl_b32_shapes.txt
timeout 600 python tools/latency_b1.py vitl vitb > gpurun_out/r5e/latency_b1.txt 2>&1
cat gpurun_out/r5e/latency_b1.txt
timeout 900 python tools/run_configs.py > gpurun_out/r5e/other_configs.txt 2>&1
cat gpurun_out/r5e/other_configs.txt
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k "reload_repacks" -p no:cacheprovider 2>&1 | tail -2
