#!/bin/bash
# round 5, session r: the fp8 policy as shipped (raw models: encoder + head; sigmoid ladder: second rung; 'ssi' and first rung: fp16 terms) -- ubench, full suite, parity table, configs, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5r
O=gpurun_out/r5r
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 tools/ubench/mfma_f8_corr > $O/mfma_f8_corr.txt 2>&1; tail -n 8 $O/mfma_f8_corr.txt
( time timeout 1800 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -n 30 ) > $O/gpu_suite.txt 2>&1; tail -n 6 $O/gpu_suite.txt
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden or batch32" -p no:cacheprovider 2>&1 | grep "rel-L1" | sed 's/^\.//' > $O/parity.txt; sort -t= -k2 -g $O/parity.txt | tail -n 5
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu > $O/configs.txt; cat $O/configs.txt
timeout 900 python bench.py --no-cpu-baseline --no-traffic > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5r/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["rel_l1"], d.get("rel_l1_low_mean"), d.get("low_mean", {}).get("ms_per_step"), d["precision_ladder"])
PY
