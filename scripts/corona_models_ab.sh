#!/bin/bash
# corona -> disc, 10⁶ sky samples of a ring corona (one position, far from the axis) and a disc corona (a position per sample): the rays
# dealt by predicted cost against sample order (knob sky_deal)
for model in ring disc lamp; do for kn in "" "sky_deal=0" "" "sky_deal=0"; do
  echo "corona $model [$kn] $(CORONA_MODEL=$model SIB_KNOBS=$kn timeout 300 python scripts/sibling_workloads.py corona 6 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel', round(d['ms_median_after_warmup'],3), 'call', round(d['call_ms_median_after_warmup'],3), 'steps/ray', round(d['steps_per_ray'],1), 'bins', d['finite_bins'])")"
done; done
