#!/bin/bash
# corona -> disc, 10⁶ sky samples (scripts/sibling_workloads.py corona): the rays dealt by predicted cost (working tree) against
# the rays dealt by azimuth within chunks (abv/deal_r6z.so = the library of commit 0b8c10d: AB_DIR=abv scripts/build_variant.sh 0b8c10d
# deal_r6z -- abv/ travels to the GPU box, ab/ does not), interleaved on one box
run() { echo "corona [$1] [$2] $(GRADUS_MI355X_LIB=$1 SIB_KNOBS=$2 python scripts/sibling_workloads.py corona 8 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel', d['ms_median_after_warmup'], sorted(d['ms'])[:3], 'call', d['call_ms_median_after_warmup'], 'bins', d['finite_bins'])")"; }
for rep in 1 2 3; do
  run "" ""
  run abv/deal_r6z.so ""
done
run "" "sky_deal=0"
run "" "kernel=1"
run "" "kernel=0,block=64"
