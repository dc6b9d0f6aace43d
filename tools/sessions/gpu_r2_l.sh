#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2l
O=gpurun_out/r2l
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "forced_tile" 2>&1 | tail -n 6 | tee $O/test_tiles.log
for v in 4 16 4 16; do ADA_IGEMM_VARIANT=$v REPS=20 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | sed "s/^SCHED.*/variant $v/" | tee -a $O/gemm_ab.txt; done
for v in 4 16 4 16; do
  echo "gemm variant $v" | tee -a $O/bench_ab.txt
  ADA_IGEMM_VARIANT=$v timeout 600 python bench.py --no-cpu-baseline --steps 15 --warmup 4 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), 'igemm TF', round(d['roofline']['achieved'],1), 'rel_l1', d['rel_l1'])" | tee -a $O/bench_ab.txt
done
