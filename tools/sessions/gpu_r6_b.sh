#!/bin/bash
# round 6, session b: split attention output / MLP hidden ([hi | lo8 | hi8] from the attention kernel and the SwiGLU epilogue), calibration rules (global / cross),
# head branches on side streams and the two-half-batch experiment at config 2, then the GPU suite and the bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r6b
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_f8.py -m gpu -q -s -p no:cacheprovider -k "split_output" 2>&1 | grep -v amdgpu | tail -15
KS=0,8,12,16,40 F8=none timeout 900 python tools/stage_errors.py raw_vitg_224 2>&1 | grep -v amdgpu > gpurun_out/r6b/stage_errors_vitg_fp16terms.txt
tail -8 gpurun_out/r6b/stage_errors_vitg_fp16terms.txt
KS=0,8,12,16,40 timeout 900 python tools/stage_errors.py raw_vitg_224 raw_vitg_224_w1 raw_vitg_126x154_unc 2>&1 | grep -v amdgpu > gpurun_out/r6b/stage_errors_vitg_policy.txt
grep -A1 "stage\|tap3\|out" gpurun_out/r6b/stage_errors_vitg_policy.txt | tail -30
timeout 900 python tools/calibration_table.py 2>&1 | grep -v amdgpu > gpurun_out/r6b/calibration_table.txt
tail -60 gpurun_out/r6b/calibration_table.txt
for hs in 0 1; do ADA_HEAD_STREAMS=$hs ENCODER=vitb B=8 timeout 600 python tools/config_shapes.py 2>&1 | grep -v amdgpu | head -4 | tail -2; done > gpurun_out/r6b/config2_head_streams_ab.txt 2>&1
cat gpurun_out/r6b/config2_head_streams_ab.txt
ENCODER=vitb B=8 timeout 600 python tools/half_batch_streams.py 2>&1 | grep -v amdgpu > gpurun_out/r6b/config2_half_batches.txt; cat gpurun_out/r6b/config2_half_batches.txt
timeout 900 python tools/run_configs.py 2>&1 | grep -v amdgpu > gpurun_out/r6b/other_configs.txt; cat gpurun_out/r6b/other_configs.txt
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 > gpurun_out/r6b/gpu_suite.txt 2>&1
grep -v amdgpu gpurun_out/r6b/gpu_suite.txt | grep -v "UserWarning\|warnings.warn\|^$\|^tests/" | tail -40
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden or batch32" -p no:cacheprovider 2>&1 | grep "rel-L1" | sed 's/^\.//' > gpurun_out/r6b/parity_vs_reference_goldens.txt; wc -l gpurun_out/r6b/parity_vs_reference_goldens.txt
timeout 600 python bench.py --no-cpu-baseline --no-traffic > gpurun_out/r6b/bench.json 2> gpurun_out/r6b/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6b/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "rel_l1", "rel_l1_low_mean", "escalated_images_in_timed_steps")}, d["roofline"]["frac"], d["precision_ladder"]["r_threshold"], d["precision_ladder"]["r_of_timed_batch"], d["low_mean"]["ms_per_step"])
PY
