#!/bin/bash
# round 3, GPU job 15: final sources (fp32 keeps the inverse t-phi components formed first): full GPU suite, head + fp32 profiles, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 900 bash scripts/profile_pmc.sh r3o_head > $O/prof_head.log 2>&1
PROF_F32=1 PROF_CMD="scripts/sibling_workloads.py c5f32" PROF_KERNEL="gr32::" timeout 900 bash scripts/profile_pmc.sh r3o_c5f32 > $O/prof_c5f32.log 2>&1
rm -rf gpurun_out/prof_r3o_*/trace gpurun_out/prof_r3o_*/pmc?
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_r3o_*/summary.json")):
    s = json.load(open(f))
    print(f.split("/")[1], {k: (round(s[k], 4) if isinstance(s.get(k), float) else s.get(k)) for k in ("avg_ms", "clock_ghz", "valu_issue_per_4clk", "fp64_pipe_busy_nominal",
          "hbm_write_bytes_per_launch", "hbm_read_bytes_per_launch", "scratch_bytes", "valu_insts_per_wave", "source_sha16")})
PY
cp gpurun_out/prof_r3o_head/summary.json profiles/r3o_head_summary.json && echo profiles/r3o_head_summary.json > profiles/CURRENT
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench.err; tail -c 400 $O/bench_default.json; echo
