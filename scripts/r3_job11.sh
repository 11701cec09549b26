#!/bin/bash
# round 3, GPU job 11: final library (tangent objects with the default contraction): full GPU suite, tangent kernel profile
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs > $O/pytest.log 2>&1; tail -6 $O/pytest.log
PROF_CMD="scripts/sibling_workloads.py tangent" PROF_KERNEL="k_trace_lane<grt::" timeout 900 bash scripts/profile_pmc.sh r3k_tangent > $O/prof_tangent.log 2>&1
rm -rf gpurun_out/prof_r3k_tangent/trace gpurun_out/prof_r3k_tangent/pmc?
python3 -c "
import json; s=json.load(open('gpurun_out/prof_r3k_tangent/summary.json')); print({k:s.get(k) for k in ('avg_ms','valu_insts_per_wave','scratch_bytes','arch_vgpr_per_lane','valu_issue_per_4clk')})"
timeout 900 python3 __graft_entry__.py > $O/smoke.log 2>&1; tail -3 $O/smoke.log
