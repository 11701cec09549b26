#!/bin/bash
# round 3, GPU job 13: the build with the event sampling parked in LDS (default), pinned-image direct stores: full GPU suite,
# every kernel re-profiled, the boundary calls, BASELINE configurations at full size, the default bench line, soak
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 600 python3 scripts/endpoints_pinned_time.py 2048 > $O/endpoints_pinned.log 2>&1; head -7 $O/endpoints_pinned.log
timeout 2400 bash scripts/profile_all.sh r3m > $O/profile_all.log 2>&1
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_r3m_*/summary.json")):
    s = json.load(open(f))
    print(f.split("/")[1], {k: (round(s[k], 4) if isinstance(s.get(k), float) else s.get(k)) for k in ("avg_ms", "clock_ghz", "valu_issue_per_4clk", "fp64_pipe_busy_nominal",
          "valu_lane_utilization", "hbm_write_bytes_per_launch", "hbm_read_bytes_per_launch", "scratch_bytes", "arch_vgpr_per_lane", "valu_insts_per_wave")})
PY
cp gpurun_out/prof_r3m_head/summary.json profiles/r3m_head_summary.json && echo profiles/r3m_head_summary.json > profiles/CURRENT
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench.err; tail -c 900 $O/bench_default.json
timeout 600 python3 scripts/run_configs.py > $O/run_configs.log 2>&1; cat $O/run_configs.log
timeout 1200 python3 scripts/soak.py 2000 4013 > $O/soak_2000_seed4013.log 2>&1; tail -3 $O/soak_2000_seed4013.log | cut -c1-400
