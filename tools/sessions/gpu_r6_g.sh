#!/bin/bash
# round 6, session g: the calibration's shift grid in units of the logits' spread (ladder_curve) -- the suite, the calibration table, the bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r6g
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/calibration_table.py quick 2>&1 | grep -v amdgpu | tail -8
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r6g/gpu_suite.txt 2>&1; grep "passed\|failed" gpurun_out/r6g/gpu_suite.txt | tail -3; grep "^FAILED" gpurun_out/r6g/gpu_suite.txt | head
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden or batch32" -p no:cacheprovider 2>&1 | grep "rel-L1" | sed 's/^\.//' > gpurun_out/r6g/parity.txt; sort -t= -k2 -g -r gpurun_out/r6g/parity.txt | head -5
timeout 600 python bench.py --no-cpu-baseline --no-traffic --repeats 1 > gpurun_out/r6g/bench.json 2> gpurun_out/r6g/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6g/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "rel_l1", "rel_l1_low_mean", "escalated_images_in_timed_steps")}, d["roofline"]["frac"], {k: d["low_mean"].get(k) for k in ("ms_per_step", "rel_l1")})
c = d["precision_ladder"]["calibration"]; print({k: c.get(k) for k in ("eps1", "eps2", "r_global", "r_cross", "r3_global", "r3_cross", "r_installed", "r3_installed", "div_installed")}, d["precision_ladder"]["r_of_timed_batch"])
PY
