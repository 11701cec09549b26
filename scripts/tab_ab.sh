# interleaved A/B of GR_METRIC_TABULATED kernel variants (abv/*.so, scripts/build_variant.sh with METRICS=11).
# Every variant first renders 256² under a short timeout: a variant that is wrong (garbage coefficients send rays to maxiters) or
# pathologically slow is skipped instead of holding the GPU for minutes.
for v in abv/*.so; do
  echo "=== $v"
  GRADUS_MI355X_LIB=$PWD/$v timeout 90 python scripts/tabmetric_bench.py --metrics kerr --sizes 256 --reps 1 2>&1 | grep "tabulated" | cut -c1-330 > /tmp/tab_small.log
  cat /tmp/tab_small.log
  ms=$(grep "'path': 'tabulated'," /tmp/tab_small.log | sed "s/.*'kernel_ms': \([0-9.]*\).*/\1/")
  if [ -z "$ms" ] || [ "$(python3 -c "print(int(float('$ms') > 60))")" = "1" ]; then echo "   SKIPPED (small render: '$ms' ms)"; continue; fi
  GRADUS_MI355X_LIB=$PWD/$v timeout 300 python scripts/tabmetric_bench.py --metrics kerr --sizes 1024,2048 --reps 2 2>&1 | grep "tabulated" | cut -c1-330
done
