#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2final
O=gpurun_out/r2final
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 | tee $O/smoke.txt
timeout 3000 python -m pytest tests -m gpu -x -q > $O/test_all.log 2>&1; echo "all gpu tests rc=$?" | tee $O/summary.txt; tail -n 2 $O/test_all.log
timeout 900 python bench.py 2>&1 | tail -n 1 > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['rel_l1'], d['roofline']['frac'], d['roofline_attention']['frac'], d['cpu_baseline']['value'])"
