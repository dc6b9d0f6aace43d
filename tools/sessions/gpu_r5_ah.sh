#!/bin/bash
# round 5, session ah: tools/qualify_checkpoint.py on the final policy (synthetic ViT-B, heavy-tailed fill; raw ViT-S)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5ah
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/qualify_checkpoint.py --encoder vitb --tail heavy 2>&1 | grep -v amdgpu > gpurun_out/r5ah/qualify_vitb_heavy.txt; tail -n 22 gpurun_out/r5ah/qualify_vitb_heavy.txt | cut -c1-220
timeout 900 python tools/qualify_checkpoint.py --encoder vits --raw --synthetic-seed 1 --sizes 126x154 2>&1 | grep -v amdgpu > gpurun_out/r5ah/qualify_raw_vits.txt; tail -n 8 gpurun_out/r5ah/qualify_raw_vits.txt | cut -c1-220
