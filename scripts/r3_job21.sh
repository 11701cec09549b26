#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3u; mkdir -p $O
hipcc -O2 --offload-arch=gfx950 -o /tmp/hr_thp scripts/microbench/host_register_thp.hip 2>&1 | tail -2
timeout 300 /tmp/hr_thp > $O/host_register_thp.txt 2>&1; cat $O/host_register_thp.txt
