#!/bin/bash
# round 4, session f: raw ViT-G under the new default policy (first 8 encoder blocks + head groups incl. oc1), held-out fixtures w3 / w4 included; config 5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4f
O=$PWD/gpurun_out/r4f
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden and raw_vitg" 2>&1 | grep -v amdgpu > $O/pytest_vitg.txt; grep "rel-L1\|passed\|failed" $O/pytest_vitg.txt
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "config5" 2>&1 | grep -v amdgpu | grep "rel-L1\|passed\|failed"
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu | tee $O/other_configs.txt
