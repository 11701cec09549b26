# launch-shape A/B for the fp32 line-profile kernel: persistent vs one ray per lane, refill thresholds
for kn in "" "refill_threshold=8" "refill_threshold=4" "refill_threshold=32" "kernel=0,block=256" "kernel=0,block=64" "waves_per_simd=3" "waves_per_simd=2"; do
  echo "c5f32 [$kn] $(SIB_KNOBS=$kn python scripts/sibling_workloads.py c5f32 5 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_median_after_warmup'], sorted(d['ms'])[:2])")"
done
for kn in "" "kernel=0,block=256" "refill_threshold=8"; do
  echo "c5 fp64 [$kn] $(SIB_KNOBS=$kn python scripts/sibling_workloads.py c5 4 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_median_after_warmup'], sorted(d['ms'])[:2])")"
done
