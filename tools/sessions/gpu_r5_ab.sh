#!/bin/bash
# round 5, session ab: the flat-input rung of the heads without a sigmoid + every block of ViT-S split -- ladder tests, every fixture, the probe again, the fuzz file
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5ab
O=gpurun_out/r5ab
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -x -k "ladder or flat_input or projects or encoder_split" -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -n 40 > $O/ladder_tests.txt; grep "rel-L1\|r = \|token div\|passed\|failed\|Error\|assert" $O/ladder_tests.txt | head -30
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden" -p no:cacheprovider 2>&1 | grep "rel-L1\|passed\|failed" | sed 's/^\.//' > $O/parity.txt; sort -t= -k2 -g $O/parity.txt | tail -n 6; tail -n 1 $O/parity.txt; grep "zeros\|checker\|raw_vits\|vits_ssi" $O/parity.txt
timeout 1500 python tools/degenerate_inputs_unbounded.py 2>&1 | grep -v amdgpu > $O/degenerate_unbounded.txt; cut -c1-150 $O/degenerate_unbounded.txt | awk '{ if ($0 ~ /rel-L1/) print }' | sort -t= -k2 -g | tail -8; tail -n 1 $O/degenerate_unbounded.txt
timeout 2000 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_tiling.py tests/test_gpu_metrics.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -n 3
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu
