# A/B of the third wave for the latency-starved fused kernels: dilaton-axion (dual6) and Kerr-refractive
# (dual9), default build against abv/park5.so (METRICS="6 9" scripts/build_variant.sh WORK park5 -DGR_PARK_DEFAULT=5).
for S in 1024 2048; do for w in dual6 dual9; do
  for lib in default abv/park5.so; do
    if [ $lib = default ]; then unset GRADUS_MI355X_LIB; else export GRADUS_MI355X_LIB=$PWD/$lib; fi
    echo "$w $S $lib $(SIB_SIZE=$S python scripts/sibling_workloads.py $w 6 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_median_after_warmup'], sorted(d['ms'])[:3])")"
  done
done; done
