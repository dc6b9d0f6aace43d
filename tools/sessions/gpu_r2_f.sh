#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2f
O=gpurun_out/r2f
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/test_all.log 2>&1; echo "all gpu tests rc=$?" | tee $O/summary.txt
grep -E "rel-L1|passed|failed|Error" $O/test_all.log | tail -n 40
# power while the attention kernel loops (is the chip at its cap?)
(for i in $(seq 1 12); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | head -n 4; sleep 0.5; done) > $O/power_attn.txt 2>&1 &
VARIANT=3 REPS=20000 timeout 60 python tools/bench_attn.py 2>&1 | tail -n 1 | tee -a $O/summary.txt
wait
tail -n 12 $O/power_attn.txt
for av in 3 5; do for gv in 4 8; do
  echo "attention variant $av, gemm variant $gv" | tee -a $O/bench_ab.txt
  ADA_ATTN_VARIANT=$av ADA_IGEMM_VARIANT=$gv timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_attention']['achieved'], d['roofline_attention']['avg_launch_ms'])" | tee -a $O/bench_ab.txt
done; done
