#!/bin/bash
# round 3, GPU job 27: every kernel of DESIGN.md §5's table profiled on the final sources (one hash for all summaries)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 2400 bash scripts/profile_all.sh r3z > $O/profile_all.log 2>&1
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_r3z_*/summary.json")):
    s = json.load(open(f))
    print(f.split("/")[1], {k: (round(s[k], 4) if isinstance(s.get(k), float) else s.get(k)) for k in ("avg_ms", "clock_ghz", "valu_issue_per_4clk", "fp64_pipe_busy_nominal",
          "valu_lane_utilization", "hbm_write_bytes_per_launch", "hbm_read_bytes_per_launch", "scratch_bytes", "arch_vgpr_per_lane", "source_sha16")})
PY
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench.err; tail -c 300 $O/bench_default.json; echo
