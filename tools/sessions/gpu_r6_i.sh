#!/bin/bash
# round 6, session i: the 192-channel output_conv2 of the raw ViT-G head padded to 256 channels so that its correction terms take the fp8 pipe: ViT-G fixtures, config 5 A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r6i
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "raw_vitg or config5" -p no:cacheprovider 2>&1 | grep "rel-L1\|passed\|failed" | sed 's/^\.//' | tail -22
for p in 0 1; do echo "ADA_OC2_PAD128=$p"; ADA_OC2_PAD128=$p timeout 900 python tools/run_configs.py 2>&1 | grep "^config 5" | cut -c1-160; done
