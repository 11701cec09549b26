#!/bin/bash
# round 3, GPU job 17: one vs two renders in flight at N = 1; smoke()
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3q; mkdir -p $O
for s in 1 2 1 2; do timeout 300 python3 bench.py --streams $s --no-cpu-baseline --no-host-call --steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('streams', d['config'].get('renders_in_flight'), 'ms_per_step', round(d['ms_per_step'],3), 'value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'launch_ms', d['roofline']['launch_ms'])" | tee -a $O/streams.txt; done
timeout 600 python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; echo smoke rc=$?; tail -2 $O/smoke.log
