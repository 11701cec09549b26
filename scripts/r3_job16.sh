#!/bin/bash
# round 3, GPU job 16: 100 000 random scenes on the final build against the oracle
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3p; mkdir -p $O
timeout 3000 python3 scripts/soak.py 100000 31337 > $O/soak_100000_seed31337.log 2>&1; head -1 $O/soak_100000_seed31337.log | cut -c1-500
