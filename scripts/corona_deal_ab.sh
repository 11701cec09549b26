#!/bin/bash
# corona -> disc, 10⁶ sky samples (scripts/sibling_workloads.py corona): launch shapes with the rays dealt by direction
for kn in "" "sky_deal=0" "" "sky_deal=0" "kernel=0,block=256" "kernel=1" "kernel=0,block=256,sky_deal=0" "kernel=0,sky_deal=0"; do
  echo "corona [$kn] $(SIB_KNOBS=$kn python scripts/sibling_workloads.py corona 6 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_median_after_warmup'], sorted(d['ms'])[:3])")"
done
