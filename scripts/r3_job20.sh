#!/bin/bash
# round 3, GPU job 20: pinned-block pool: what a caller sees per call; GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3t; mkdir -p $O
timeout 600 python3 scripts/endpoints_call_time.py 2048 > $O/endpoints_call_time.txt 2>&1; cat $O/endpoints_call_time.txt
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs > $O/pytest.log 2>&1; tail -3 $O/pytest.log
