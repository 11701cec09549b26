#!/bin/bash
# round 3, GPU job 5: the boundary proper into pinned / pageable memory; BASELINE configurations at full size; soak
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3e; mkdir -p $O
timeout 600 python3 scripts/endpoints_pinned_time.py 2048 > $O/endpoints_pinned.log 2>&1; head -4 $O/endpoints_pinned.log
timeout 600 python3 scripts/run_configs.py > $O/run_configs.log 2>&1; cat $O/run_configs.log
timeout 1500 python3 scripts/soak.py 6000 3003 > $O/soak_6000_seed3003.log 2>&1; tail -12 $O/soak_6000_seed3003.log
