#!/bin/bash
# HIP-graph replay for small batches: latency A/B, then the whole GPU suite, smoke and the default bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2q; mkdir -p $O
timeout 900 python tools/latency_b1.py vitl vitb vits 2>&1 | grep -v amdgpu.ids | tee $O/latency_graph_ab.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 | tee $O/smoke.txt
timeout 3000 python -m pytest tests -m gpu -x -q > $O/test_all.log 2>&1; echo "all gpu tests rc=$?" | tee $O/summary.txt; tail -n 3 $O/test_all.log
timeout 900 python bench.py 2>&1 | tail -n 1 > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['rel_l1'], d['roofline']['frac'], d['roofline_attention']['frac'], d['cpu_baseline']['value'])"
