#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3s; mkdir -p $O
timeout 300 python3 scripts/host_alloc_time.py 608 > $O/host_alloc_time.txt 2>&1; cat $O/host_alloc_time.txt
