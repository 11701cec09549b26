#!/bin/bash
# GR_METRIC_TABULATED: degree x grid A/B (abv/deg{5,6}.so from scripts/build_variant.sh with METRICS=11 -DGR_TAB_DEGREE=..., the
# in-tree library = degree 7), C2 scene at 1024² and the headline scene at 2048², each against the fused Kerr kernel: time, image
# error, table size.     scripts/tab_degree_ab.sh > gpurun_out/tab_degree_ab.log
run() {  # lib grid
  echo "=== $1 grid $2"
  if [ "$1" = "deg7" ]; then unset GRADUS_MI355X_LIB; else export GRADUS_MI355X_LIB=$PWD/abv/$1.so; fi
  timeout 400 python scripts/tabmetric_bench.py --metrics kerr --sizes 1024,2048 --reps 2 --grid $2 2>&1 | grep "tabulated\|Error\|error" | cut -c1-420
}
run deg7 8,32
run deg6 8,32
run deg6 12,48
run deg5 12,48
run deg5 16,64
run deg5 24,96
