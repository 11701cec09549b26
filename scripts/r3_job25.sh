#!/bin/bash
# round 3, GPU job 25: head-kernel profile and default bench line on the final sources (the hash covers the host unit too)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 900 bash scripts/profile_pmc.sh r3y_head > $O/prof_head.log 2>&1
rm -rf gpurun_out/prof_r3y_head/trace gpurun_out/prof_r3y_head/pmc?
cp gpurun_out/prof_r3y_head/summary.json profiles/r3y_head_summary.json && echo profiles/r3y_head_summary.json > profiles/CURRENT
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r3y/bench_default.json").read().strip().split("\n")[-1]); r=d["roofline"]
print({k:d[k] for k in ("value","ms_per_step","value_host_call")}, r["frac"], r["frac_of_sustained_fma_stream"], (r.get("fp64_issue") or {}).get("frac"), r["traffic"], r["executed"].get("profile"), r["executed"].get("profile_error"))
s=json.load(open("gpurun_out/prof_r3y_head/summary.json")); print(s["avg_ms"], s["clock_ghz"], s["source_sha16"], s["valu_insts_per_wave"])
PY
