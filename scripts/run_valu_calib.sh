#!/bin/bash
# The calibration behind the utilisation numbers of profiles/*_summary.json (profiles/r3_valu_calib.json): builds
# scripts/microbench/valu_calib FROM ITS SOURCE (the binary is not committed), runs it bare and under the counters
# (their own pass, no tracing domain), and folds the result with scripts/summarize_calib.py.
#   scripts/run_valu_calib.sh [tag=calib]   ->  gpurun_out/<tag>/calib_summary.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
T=${1:-calib}; O=$R/gpurun_out/$T; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/microbench/valu_calib scripts/microbench/valu_calib.hip
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 300 ./scripts/microbench/valu_calib 3 25 > $O/calib_bare.jsonl 2>&1
# the program itself follows `--` (no env / bash -c hop: the profiler initialises the GPU first)
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $O/calib_pmc -- ./scripts/microbench/valu_calib 3 25 > $O/calib_pmc.log 2>&1
python3 scripts/summarize_calib.py $O/calib_pmc $O/calib_bare.jsonl > $O/calib_summary.txt 2>&1
find $O/calib_pmc -name '*.csv' -size +2M -delete
tail -40 $O/calib_summary.txt
