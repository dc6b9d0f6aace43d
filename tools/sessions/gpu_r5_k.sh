#!/bin/bash
# round 5, session k: fabric traffic of every igemm launch of the ViT-L bs=32 forward against its algorithmic bytes (where roofline.traffic's 1.6x comes from)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5k
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && timeout 1500 python3 $GRAFT_REPO_ROOT/tools/pmc_traffic_per_shape.py > $GRAFT_REPO_ROOT/gpurun_out/r5k/traffic_per_shape.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r5k/traffic_per_shape.txt | tail -45
