#!/bin/bash
# round 6, session e: sub-pixel merge and commuted output_conv1 under split operands (the second rung, raw ViT-B / ViT-L heads), second-rung-first: kernel test,
# ladder tests, every fixture, the low-mean twin with and without the two restructurings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r6e
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_f8.py -m gpu -q -s -p no:cacheprovider -k "sub_pixel" 2>&1 | grep -v amdgpu | grep -v "^$" | tail -12
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -p no:cacheprovider -k "second_rung or ladder or flat_input or class_tokens" 2>&1 | grep -v amdgpu | grep -v "^$\|Warning\|warnings.warn\|^tests/" | tail -15
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden or batch32" -p no:cacheprovider 2>&1 | grep "rel-L1\|passed\|failed" | sed 's/^\.//' > gpurun_out/r6e/parity.txt; sort -t= -k2 -g -r gpurun_out/r6e/parity.txt | head -8; tail -1 gpurun_out/r6e/parity.txt
for sw in 1 0; do ADA_OC1_COMMUTE_SPLIT=$sw ADA_SUBPIXEL_SPLIT=$sw timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-kernel-timer --repeats 1 > gpurun_out/r6e/bench_split_restructurings$sw.json 2> gpurun_out/r6e/bench$sw.err; python - <<PY
import json
d = json.loads(open("gpurun_out/r6e/bench_split_restructurings$sw.json").read().strip().splitlines()[-1])
print("split restructurings = $sw", {k: d.get(k) for k in ("value", "ms_per_step", "rel_l1")}, {k: d["low_mean"].get(k) for k in ("ms_per_step", "images_per_sec", "escalated_images_per_step", "second_rung_first_calls", "rel_l1")})
PY
done
timeout 900 python tools/run_configs.py 2>&1 | grep "^config" | cut -c1-200
