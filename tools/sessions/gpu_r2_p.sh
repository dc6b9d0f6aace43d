#!/bin/bash
# board power / sclk while each dominant kernel loops alone
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2p
O=gpurun_out/r2p
sample() { for i in $(seq 1 10); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | tr '\n' ' '; echo; sleep 0.4; done; }
for shape in qkv fc1 fc2 "K=8192"; do
  echo "== igemm $shape" | tee -a $O/power.txt
  (ONLY="$shape" REPS=12000 timeout 40 python tools/bench_gemm.py > $O/gemm_$$.log 2>&1 &) ; sleep 8; sample | tee -a $O/power.txt; wait; sleep 2
done
echo "== attention (mix)" | tee -a $O/power.txt
(VARIANT=5 REPS=20000 timeout 40 python tools/bench_attn.py > /dev/null 2>&1 &); sleep 8; sample | tee -a $O/power.txt; wait
echo "== whole forward (bench loop)" | tee -a $O/power.txt
(timeout 60 python bench.py --no-cpu-baseline --steps 400 --warmup 3 > /dev/null 2>&1 &); sleep 15; sample | tee -a $O/power.txt; wait
