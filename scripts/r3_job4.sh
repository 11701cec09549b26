#!/bin/bash
# round 3, GPU job 4: full GPU suite; A/B (uniform constants in SGPRs, cold lane storage); head-kernel profile with and
# without the cold store (scratch attribution); every sibling kernel; the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs > $O/pytest.log 2>&1; tail -12 $O/pytest.log
timeout 600 python3 scripts/ab_bench.py abv/k0.so abv/knouni.so abv/kcold.so --rounds 12 > $O/ab_kerr.log 2>&1; cat $O/ab_kerr.log
timeout 2400 bash scripts/profile_all.sh r3d > $O/profile_all.log 2>&1
export GRADUS_MI355X_LIB=$R/abv/kcold.so
timeout 600 bash scripts/profile_pmc.sh r3d_head_coldstore > $O/prof_cold.log 2>&1
unset GRADUS_MI355X_LIB
rm -rf gpurun_out/prof_r3d_head_coldstore/trace gpurun_out/prof_r3d_head_coldstore/pmc?
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_r3d_*/summary.json")):
    s = json.load(open(f))
    print(f.split("/")[1], {k: (round(s[k], 4) if isinstance(s.get(k), float) else s.get(k)) for k in ("avg_ms", "clock_ghz", "valu_issue_per_4clk", "fp64_pipe_busy_nominal",
          "valu_lane_utilization", "hbm_write_bytes_per_launch", "scratch_bytes", "arch_vgpr_per_lane")})
PY
cp gpurun_out/prof_r3d_head/summary.json profiles/r3d_head_summary.json && echo profiles/r3d_head_summary.json > profiles/CURRENT
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench.err; tail -c 600 $O/bench_default.json
