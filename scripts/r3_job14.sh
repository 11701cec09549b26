#!/bin/bash
# round 3, GPU job 14: which right-hand-side change moved the fp32 kernels' flagged-ray fraction
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3n; mkdir -p $O
for v in f0 fwis ftphi far fw1 fall; do
  GRADUS_MI355X_LIB=$R/abv/$v.so timeout 300 python3 scripts/fp32_flag_ab.py >> $O/fp32_flag_ab.txt 2>> $O/err.log
done
cat $O/fp32_flag_ab.txt
