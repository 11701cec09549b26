#!/bin/bash
# Collect rocprofv3 kernel-trace stats and PMC passes (separate runs: counters are never combined with tracing
# domains) for one command, then fold them into gpurun_out/prof_<tag>/summary.json.
#   scripts/profile_pmc.sh <tag> [bench args...]                       -> bench.py (the 2048² Kerr headline kernel)
#   PROF_CMD="scripts/sibling_workloads.py c4" PROF_KERNEL=k_trace scripts/profile_pmc.sh <tag>
# The program after `--` is always python3 itself (no env / bash -c hop: the profiler initialises the GPU first).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
WARM=${PROF_WARMUP:-2}
if [ -n "$PROF_CMD" ]; then CMD="$PROF_CMD"; else CMD="bench.py --steps 12 --warmup $WARM --no-cpu-baseline --no-host-call --no-configs $@"; fi
NEEDLE=${PROF_KERNEL:-k_trace}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $CMD > $OUT/bench.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcA -- python3 $CMD > /dev/null 2>&1
if [ "${PROF_F32:-0}" = "1" ]; then
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmcB -- python3 $CMD > /dev/null 2>&1
else
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmcB -- python3 $CMD > /dev/null 2>&1
fi
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcC -- python3 $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcD -- python3 $CMD > /dev/null 2>&1
python3 scripts/summarize_pmc.py $OUT "$NEEDLE" $WARM
# keep what gets committed small: the stats CSV and the summary
find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
