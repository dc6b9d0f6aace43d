#!/bin/bash
# round 5, session j: BASELINE config 5 (raw ViT-G, 8 x 1022 x 1022) launch by launch with a tile sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5j
export HSA_ENABLE_IPC_MODE_LEGACY=0
RAW=1 ENCODER=vitg B=8 SIZE=1022 SWEEP=1 REPS=3 timeout 2400 python tools/config_shapes.py > gpurun_out/r5j/config5_shapes.txt 2>&1
head -5 gpurun_out/r5j/config5_shapes.txt
