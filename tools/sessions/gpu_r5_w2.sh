#!/bin/bash
# round 5, session w2: constant / checkerboard inputs on the third rung -- ladder tests, every fixture, the output-range sweep (default and held-out seed)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5w2
O=gpurun_out/r5w2
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -x -k "ladder or projects" -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -n 40 > $O/ladder_tests.txt; grep "rel-L1\|r = \|passed\|failed\|Error\|assert" $O/ladder_tests.txt | head -30
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden" -p no:cacheprovider 2>&1 | grep "rel-L1\|passed\|failed" | sed 's/^\.//' > $O/parity.txt; sort -t= -k2 -g $O/parity.txt | tail -n 5; tail -n 1 $O/parity.txt; grep "zeros\|checker" $O/parity.txt
ADA_FUZZ_SCALE=3 ADA_FUZZ_SEED=11 timeout 2700 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s -p no:cacheprovider -k "across_the_output_range" 2>&1 | grep -E "rel-L1|passed|failed|Error" > $O/fuzz_range.txt
grep "rel-L1" $O/fuzz_range.txt | sed 's/.*rel-L1[^=]*= *//' | sort -g | tail -4; grep -E "passed|failed" $O/fuzz_range.txt; grep -c "third rung" $O/fuzz_range.txt
