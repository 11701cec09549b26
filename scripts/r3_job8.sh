#!/bin/bash
# round 3, GPU job 8: RHS trims (shared r²+a², uniform Johannsen factors) A/B against the previous commit; cold lane storage
# around the event sampling only (GR_COLD_LDS=2): time, HBM write bytes, parity
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 600 python3 scripts/ab_bench.py abv/jold.so abv/w0.so abv/c2.so abv/c1.so --rounds 12 > $O/ab_kerr.log 2>&1; cat $O/ab_kerr.log
timeout 600 python3 scripts/ab_bench.py abv/jold.so abv/w0.so --rounds 10 --size 1024 --workload johannsen > $O/ab_johannsen.log 2>&1; cat $O/ab_johannsen.log
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs > $O/pytest.log 2>&1; tail -4 $O/pytest.log
export GRADUS_MI355X_LIB=$R/abv/c2.so
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs > $O/pytest_c2.log 2>&1; tail -8 $O/pytest_c2.log
timeout 600 bash scripts/profile_pmc.sh r3h_head_cold2 > $O/prof_cold2.log 2>&1
unset GRADUS_MI355X_LIB
timeout 600 bash scripts/profile_pmc.sh r3h_head > $O/prof_head.log 2>&1
rm -rf gpurun_out/prof_r3h_*/trace gpurun_out/prof_r3h_*/pmc?
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_r3h_*/summary.json")):
    s = json.load(open(f))
    print(f.split("/")[1], {k: (round(s[k], 4) if isinstance(s.get(k), float) else s.get(k)) for k in ("avg_ms", "clock_ghz", "valu_issue_per_4clk", "fp64_pipe_busy_nominal",
          "valu_lane_utilization", "hbm_write_bytes_per_launch", "hbm_read_bytes_per_launch", "scratch_bytes", "lds_bytes", "valu_insts_per_wave")})
PY
