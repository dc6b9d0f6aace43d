#!/bin/bash
# end-of-round validation: smoke, whole GPU suite, default bench line, bench through the RCCL path with one rank
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2final2; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 | tee $O/smoke.txt
timeout 3000 python -m pytest tests -m gpu -x -q > $O/test_all.log 2>&1; echo "all gpu tests rc=$?" | tee $O/summary.txt; tail -n 3 $O/test_all.log
timeout 900 python bench.py 2>&1 | tail -n 1 > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['rel_l1'], d['roofline']['frac'], d['roofline_attention']['frac'], d['cpu_baseline']['value'])"
ADA_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 900 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -n 1 > $O/bench_dist1.json; python -c "
import json; d=json.load(open('$O/bench_dist1.json')); print('rccl path, 1 rank:', d['value'], d['ms_per_step'], d['n_gpus'])"
