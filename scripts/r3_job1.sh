#!/bin/bash
# round 3, GPU job 1: (1) calibration streams bare and under the counters, (2) interleaved A/B of the head-kernel variants
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 300 ./scripts/microbench/valu_calib 3 25 > $O/calib_bare.jsonl 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $O/calib_pmc -- ./scripts/microbench/valu_calib 3 25 > $O/calib_pmc.log 2>&1
python3 scripts/summarize_calib.py $O/calib_pmc $O/calib_bare.jsonl > $O/calib_summary.txt 2>&1
find $O/calib_pmc -name '*.csv' -size +2M -delete
timeout 900 python3 scripts/ab_bench.py abv/base.so abv/v1.so abv/v2.so abv/v3.so abv/v4.so --rounds 10 > $O/ab.log 2>&1
cat $O/calib_summary.txt | tail -80
cat $O/ab.log
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
