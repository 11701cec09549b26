#!/bin/bash
# Collect rocprofv3 kernel-trace stats and PMC passes for the bench workload.
# usage: scripts/profile_pmc.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcA -- python3 bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmcB -- python3 bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcC -- python3 bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcD -- python3 bench.py $ARGS > /dev/null 2>&1
python3 scripts/summarize_pmc.py $OUT
