#!/bin/bash
# round 4, session ad: new reference fixtures generated after the precision policy was fixed (raw ViT-B / ViT-L at 518^2; then the ssi head on ViT-B / ViT-L)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ad
O=$PWD/gpurun_out/r4ad
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -s -k "714x1022_heavy" 2>&1 | grep "rel-L1\|passed\|failed" | tee $O/new_fixtures.txt
