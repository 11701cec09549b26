#!/bin/bash
# round 3, GPU job 2: GPU suite on the new defaults; head-kernel profile (8 x 8 and 16 x 4 tiles); bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1200 python3 -m pytest tests -m gpu -x -q -rxX > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 600 bash scripts/profile_pmc.sh r3b_head > $O/prof_head.log 2>&1
timeout 600 bash scripts/profile_pmc.sh r3b_head_t16 --tile-rows 16 > $O/prof_head_t16.log 2>&1
for t in r3b_head r3b_head_t16; do find gpurun_out/prof_$t -name '*.csv' -size +1M -delete; done
python3 - <<'PY'
import json
for t in ("r3b_head", "r3b_head_t16"):
    try:
        s = json.load(open(f"gpurun_out/prof_{t}/summary.json"))
        print(t, {k: s.get(k) for k in ("avg_ms", "min_ms", "clock_ghz", "valu_issue_per_4clk", "fp64_pipe_busy_nominal", "valu_insts_per_wave",
                                        "hbm_write_bytes_per_launch", "hbm_read_bytes_per_launch", "scratch_bytes", "arch_vgpr_per_lane", "executed_fp64_flops_per_ray")})
    except Exception as e:
        print(t, "failed", e)
PY
timeout 600 python3 bench.py --steps 40 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
