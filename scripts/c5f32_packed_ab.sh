#!/bin/bash
# fp32 kernels: packed sums (v_pk_fma_f32 on the stage arguments, new state and error estimate; the in-tree build, GR_PK_F32 = 1)
# against the scalar sums (abv/nopk.so = scripts/build_variant.sh WORK nopk -DGR_PK_F32=0 with METRICS=0), interleaved: C5's fp32
# sweep point (4096² rays, tol 1e-5) and, as a control, the fp64 line profile, which the switch does not touch.
for rep in 1 2 3; do
  for v in packed scalar; do
    if [ "$v" = "packed" ]; then unset GRADUS_MI355X_LIB; else export GRADUS_MI355X_LIB=$PWD/abv/nopk.so; fi
    echo "c5f32 $v rep $rep: $(python scripts/sibling_workloads.py c5f32 5 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_median_after_warmup'], sorted(d['ms'])[:3])")"
  done
done
unset GRADUS_MI355X_LIB
echo "c5 fp64 (control): $(python scripts/sibling_workloads.py c5 4 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_median_after_warmup'], sorted(d['ms'])[:2])")"
