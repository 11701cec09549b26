#!/bin/bash
# round 3, GPU job 7: wave-transposed end-point stores + direct stores into pinned host memory; full GPU suite; every kernel
# re-profiled (kernel trace + PMC) on the new build; the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs -s > $O/pytest.log 2>&1; grep -E "end points|passed|failed|Error" $O/pytest.log | tail -12
timeout 600 python3 scripts/endpoints_pinned_time.py 2048 > $O/endpoints_pinned.log 2>&1; head -7 $O/endpoints_pinned.log
timeout 2400 bash scripts/profile_all.sh r3g > $O/profile_all.log 2>&1
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_r3g_*/summary.json")):
    s = json.load(open(f))
    print(f.split("/")[1], {k: (round(s[k], 4) if isinstance(s.get(k), float) else s.get(k)) for k in ("avg_ms", "clock_ghz", "valu_issue_per_4clk", "fp64_pipe_busy_nominal",
          "valu_lane_utilization", "hbm_write_bytes_per_launch", "scratch_bytes", "arch_vgpr_per_lane")})
PY
cp gpurun_out/prof_r3g_head/summary.json profiles/r3g_head_summary.json && echo profiles/r3g_head_summary.json > profiles/CURRENT
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench.err; tail -c 600 $O/bench_default.json
