#!/bin/bash
# round 6, session d: the second rung's head first after a call whose images left the first rung (DepthEngine._second_rung_first): bit-identity tests, the ladder tests,
# the low-mean twin of the timed batch with and without it
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r6d
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -p no:cacheprovider -k "second_rung or ladder or flat_input or class_tokens or f8_terms" 2>&1 | grep -v amdgpu | grep -v "^$\|Warning\|warnings.warn" | tail -25
timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k "output_range or hostile" 2>&1 | grep -v amdgpu | tail -3
for st in 0 1; do ADA_LADDER_STICKY=$st timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-kernel-timer --repeats 1 > gpurun_out/r6d/bench_sticky$st.json 2> gpurun_out/r6d/bench_sticky$st.err; python - <<PY
import json
d = json.loads(open("gpurun_out/r6d/bench_sticky$st.json").read().strip().splitlines()[-1])
print("ADA_LADDER_STICKY=$st", {k: d.get(k) for k in ("value", "ms_per_step", "rel_l1", "rel_l1_low_mean")}, {k: d["low_mean"].get(k) for k in ("ms_per_step", "images_per_sec", "escalated_images_per_step", "second_rung_first_calls", "rel_l1")})
PY
done
