#!/bin/bash
# round 3, GPU job 28: two-term rotation polynomials at the interior stage points: GPU suite, every kernel profiled, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r4a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 2400 bash scripts/profile_all.sh r4a > $O/profile_all.log 2>&1
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_r4a_*/summary.json")):
    s = json.load(open(f))
    print(f.split("/")[1], {k: (round(s[k], 4) if isinstance(s.get(k), float) else s.get(k)) for k in ("avg_ms", "clock_ghz", "valu_issue_per_4clk", "fp64_pipe_busy_nominal",
          "hbm_write_bytes_per_launch", "hbm_read_bytes_per_launch", "scratch_bytes", "arch_vgpr_per_lane", "valu_insts_per_wave", "source_sha16")})
PY
cp gpurun_out/prof_r4a_head/summary.json profiles/r4a_head_summary.json && echo profiles/r4a_head_summary.json > profiles/CURRENT
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4a/bench_default.json").read().strip().split("\n")[-1]); r=d["roofline"]
print({k:d[k] for k in ("value","ms_per_step","value_host_call")}, r["frac"], r["frac_of_sustained_fma_stream"], (r.get("fp64_issue") or {}).get("frac"), r["traffic"], r["executed"].get("profile"))
PY
timeout 600 python3 scripts/endpoints_pinned_time.py 2048 > $O/endpoints_pinned.log 2>&1; head -5 $O/endpoints_pinned.log
timeout 600 python3 scripts/run_configs.py > $O/run_configs.log 2>&1; grep "^C[1-4]" $O/run_configs.log | cut -c1-200
