#!/bin/bash
# round 5, session t: A/B -- the first rung's four 1x1 projects as a full split product with fp8 correction terms (ADA_RUNG1_PROJ_F8=1) on every sigmoid ViT-B / ViT-L fixture; what it costs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5t
O=gpurun_out/r5t
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 0 1; do
  ADA_RUNG1_PROJ_F8=$v timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden and (vitb or vitl) and not raw and not ssi" -p no:cacheprovider 2>&1 | grep "rel-L1\|passed\|failed" | sed 's/^\.//' > $O/parity_$v.txt; tail -n 1 $O/parity_$v.txt
done
for v in 0 1 0 1; do
  ADA_RUNG1_PROJ_F8=$v timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-low-mean --repeats 0 > $O/bench_$v.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("gpurun_out/r5t/bench_$v.json").read().strip().splitlines()[-1])
print("ADA_RUNG1_PROJ_F8=$v", d["value"], d["ms_per_step"], d["rel_l1"])
PY
done
ADA_RUNG1_PROJ_F8=1 timeout 600 python tools/run_configs.py 2>&1 | grep "config 2"
ADA_RUNG1_PROJ_F8=0 timeout 600 python tools/run_configs.py 2>&1 | grep "config 2"
