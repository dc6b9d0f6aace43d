#!/bin/bash
# round 4, session x: where the GPU suite's wall time goes (all durations >= 0.5 s)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4x
O=$PWD/gpurun_out/r4x
timeout 1500 python -m pytest tests -q -m gpu --durations=0 --durations-min=0.4 2>&1 | grep -v "^\.\|^$" > $O/durations.txt
tail -3 $O/durations.txt
