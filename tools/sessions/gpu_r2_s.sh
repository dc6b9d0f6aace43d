#!/bin/bash
# exp2-of-polynomial GELU and rcp SiLU: kernel tests, whole suite, per-shape table, bench x2.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gelu or swiglu or fold" 2>&1 | tail -n 3 | tee $O/kernel_tests.txt
timeout 600 python tools/bench_shapes.py --reps 10 2>&1 | grep -v amdgpu.ids | head -n 12 | tee $O/shapes_top.txt
timeout 3000 python -m pytest tests -m gpu -x -q > $O/test_all.log 2>&1; echo "all gpu tests rc=$?" | tee $O/summary.txt; tail -n 3 $O/test_all.log
grep -h "rel-L1" $O/test_all.log | head -n 5
for i in 1 2; do timeout 900 python bench.py 2>&1 | tail -n 1 > $O/bench_$i.json; python -c "
import json; d=json.load(open('$O/bench_$i.json')); print(d['value'], d['ms_per_step'], d['rel_l1'], d['roofline']['frac'], d['roofline_attention']['frac'])"; done
timeout 600 python -m pytest tests/test_gpu_model.py -q -s -k "golden" 2>&1 | grep "rel-L1" | tee $O/parity.txt
