#!/bin/bash
# round 3, GPU job 18: per-rank device work of a sharded render on the final build (one rank's shard on one GPU, no RCCL)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3r; mkdir -p $O
for spec in 2:0 2:1 4:0 4:1 4:2 4:3 8:0 8:1 8:2 8:3 8:4 8:5 8:6 8:7; do
  for s in 1 2; do
    timeout 200 python3 bench.py --emulate-shard $spec --streams $s --steps 200 --no-cpu-baseline --no-host-call 2>/dev/null | tail -1 >> $O/emulated_shards.jsonl
  done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r3r/emulated_shards.jsonl"):
    d = json.loads(l); print(d["emulated_shard"], d["renders_in_flight"], round(d["ms_per_step"], 3), d["rays"])
PY
