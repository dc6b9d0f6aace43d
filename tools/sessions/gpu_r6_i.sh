#!/bin/bash
# round 6, session i: the rung most of the previous call's images ended on runs first -- third-rung streams (raw maps that are mostly clipped; the bf16 build): tests,
# config 5's un-centred twin, the bf16 library at the headline configuration
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r6i
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -p no:cacheprovider -k "rung_first or ladder or flat_input or class_tokens" 2>&1 | grep -v amdgpu | grep -v "^$\|Warning\|warnings.warn\|^tests/" | tail -14
timeout 900 python tools/run_configs.py 2>&1 | grep "^config 5" | cut -c1-220
ADA_HIP_LIB=$PWD/amodal-depth-anything_amd/csrc/libada_hip_bf16.so timeout 900 python bench.py --no-cpu-baseline --no-traffic --no-kernel-timer --repeats 1 --steps 10 --warmup 3 > gpurun_out/r6i/bf16_bench.json 2> gpurun_out/r6i/bf16_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6i/bf16_bench.json").read().strip().splitlines()[-1])
print("bf16 operands:", {k: d.get(k) for k in ("value", "ms_per_step", "dtype", "rel_l1", "rel_l1_low_mean", "escalated_images_in_timed_steps")})
PY
