#!/bin/bash
# round 4, session ac: board power and shader clock while the whole forward loops (final kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ac
O=$PWD/gpurun_out/r4ac
timeout 300 python tools/power_loops.py --forward --seconds 20 2>/dev/null | tee $O/power_forward.txt
timeout 300 python tools/power_loops.py --seconds 8 2>/dev/null | tee -a $O/power_forward.txt
