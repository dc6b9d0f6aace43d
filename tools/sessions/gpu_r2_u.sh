#!/bin/bash
# deep residual prefetch (LDS + registers) in the fp32 residual epilogue of the 256x256 tile
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -k "igemm or tile" 2>&1 | tail -n 3 | tee $O/kernel_tests.txt
timeout 600 python tools/trace_gemm.py proj fc2 2>&1 | grep -v amdgpu.ids | tee $O/anatomy.txt
timeout 600 python tools/bench_shapes.py --reps 10 2>&1 | grep -v amdgpu.ids | head -n 8 | tee $O/shapes_top.txt
for i in 1 2; do timeout 900 python bench.py 2>&1 | tail -n 1 > $O/bench_$i.json; python -c "
import json; d=json.load(open('$O/bench_$i.json')); print(d['value'], d['ms_per_step'], d['rel_l1'], d['roofline']['frac'], d['roofline_attention']['frac'])"; done
