#!/bin/bash
# round 5, session p: fp8 correction terms in the DPT head's split groups too -- kernel tests, every reference fixture, the ladder tests, configs 2 / 5 and the low-mean twin of the bench batch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5p
O=gpurun_out/r5p
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_f8.py -q -s -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -n 60 > $O/f8_tests.txt; tail -n 30 $O/f8_tests.txt
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden or batch32 or ladder" -p no:cacheprovider 2>&1 | grep "rel-L1\|passed\|failed\|Error\|error" | sed 's/^\.//' > $O/parity_f8.txt; sort -t= -k2 -g $O/parity_f8.txt | tail -n 8; tail -n 2 $O/parity_f8.txt
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu > $O/configs_f8.txt; cat $O/configs_f8.txt
timeout 900 python bench.py --no-cpu-baseline --no-traffic > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5p/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["rel_l1"], d.get("rel_l1_low_mean"), d.get("low_mean"))
PY
ADA_F8_CORR=0 timeout 900 python bench.py --no-cpu-baseline --no-traffic --steps 5 --warmup 2 --repeats 0 > $O/bench_fp16_terms.json 2> $O/bench_fp16_terms.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5p/bench_fp16_terms.json").read().strip().splitlines()[-1])
print("fp16 terms:", d["value"], d["ms_per_step"], d["rel_l1"], d.get("rel_l1_low_mean"), d.get("low_mean"))
PY
