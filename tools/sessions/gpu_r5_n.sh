#!/bin/bash
# round 5, session n: fp8 MFMA semantics + rate for the correction terms of split-precision products (tools/ubench/mfma_f8_corr.hip)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5n
timeout 300 tools/ubench/mfma_f8_corr > gpurun_out/r5n/mfma_f8_corr.txt 2>&1
cat gpurun_out/r5n/mfma_f8_corr.txt
