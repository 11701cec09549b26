#!/bin/bash
# rocprofv3 evidence for every kernel of DESIGN.md §5's table: scripts/profile_all.sh <round tag, e.g. r2>
# -> gpurun_out/prof_<tag>_<workload>/{summary.json,kernel_stats.csv}; copy those into profiles/ to commit them.
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2}
cd $R
bash scripts/profile_pmc.sh ${T}_head
bash scripts/profile_pmc.sh ${T}_c2 --size 1024
PROF_CMD="scripts/sibling_workloads.py tabkerr" PROF_KERNEL="k_trace_lane<gr::TabulatedMetric" bash scripts/profile_pmc.sh ${T}_tabkerr
PROF_CMD="scripts/sibling_workloads.py tabc4" PROF_KERNEL="k_trace_lane<gr::TabulatedMetric" bash scripts/profile_pmc.sh ${T}_tabc4
PROF_CMD="scripts/sibling_workloads.py c4" PROF_KERNEL="k_trace_lane<gr::JohannsenMetric" bash scripts/profile_pmc.sh ${T}_c4
# the same three at 2048² (launches deep enough that their tails do not decide): the tabulated / fused ratios of DESIGN §5c
SIB_SIZE=2048 PROF_CMD="scripts/sibling_workloads.py tabkerr 6" PROF_KERNEL="k_trace_lane<gr::TabulatedMetric" bash scripts/profile_pmc.sh ${T}_tabkerr2048
SIB_SIZE=2048 PROF_CMD="scripts/sibling_workloads.py tabc4 6" PROF_KERNEL="k_trace_lane<gr::TabulatedMetric" bash scripts/profile_pmc.sh ${T}_tabc42048
SIB_SIZE=2048 PROF_CMD="scripts/sibling_workloads.py c4 6" PROF_KERNEL="k_trace_lane<gr::JohannsenMetric" bash scripts/profile_pmc.sh ${T}_c42048
PROF_CMD="scripts/sibling_workloads.py generic" PROF_KERNEL="k_trace_lane<gr::GenericMetric" bash scripts/profile_pmc.sh ${T}_generic
PROF_CMD="scripts/sibling_workloads.py c5" PROF_KERNEL="k_trace_lane<gr::KerrFamily" bash scripts/profile_pmc.sh ${T}_c5
PROF_CMD="scripts/sibling_workloads.py c5p" PROF_KERNEL="k_trace_persistent<gr::KerrFamily" bash scripts/profile_pmc.sh ${T}_c5p
PROF_F32=1 PROF_CMD="scripts/sibling_workloads.py c5f32" PROF_KERNEL="gr32::" bash scripts/profile_pmc.sh ${T}_c5f32
# BASELINE config 5 on a USER metric (Kerr through the table): fp64 kernels at 1e-9, at 1e-5, and the fp32 kernels at 1e-5
PROF_CMD="scripts/sibling_workloads.py tabc5 5" PROF_KERNEL="k_trace_lane<gr::TabulatedMetric" bash scripts/profile_pmc.sh ${T}_tabc5
PROF_CMD="scripts/sibling_workloads.py tabc5lo 5" PROF_KERNEL="k_trace_lane<gr::TabulatedMetric" bash scripts/profile_pmc.sh ${T}_tabc5lo
PROF_F32=1 PROF_CMD="scripts/sibling_workloads.py tabc5f32 5" PROF_KERNEL="gr32::" bash scripts/profile_pmc.sh ${T}_tabc5f32
PROF_CMD="scripts/sibling_workloads.py applypf" PROF_KERNEL="k_apply_pf" bash scripts/profile_pmc.sh ${T}_applypf
PROF_CMD="scripts/sibling_workloads.py endpoints" PROF_KERNEL="k_trace_lane<gr::KerrFamily" bash scripts/profile_pmc.sh ${T}_endpoints
PROF_CMD="scripts/sibling_workloads.py corona" PROF_KERNEL="k_trace" bash scripts/profile_pmc.sh ${T}_corona
PROF_CMD="scripts/sibling_workloads.py tangent" PROF_KERNEL="k_trace_lane<grt::" bash scripts/profile_pmc.sh ${T}_tangent
PROF_CMD="scripts/sibling_workloads.py tabtangent" PROF_KERNEL="k_trace_lane<grt::TabulatedMetric" bash scripts/profile_pmc.sh ${T}_tabtangent
PROF_CMD="scripts/sibling_workloads.py dual" PROF_KERNEL="k_trace_lane<gr::GenericMetricT<3>" bash scripts/profile_pmc.sh ${T}_bumblebee
PROF_CMD="scripts/sibling_workloads.py dual2" PROF_KERNEL="k_trace_lane<gr::GenericMetricT<2>" bash scripts/profile_pmc.sh ${T}_morristhorne
PROF_CMD="scripts/sibling_workloads.py dual6" PROF_KERNEL="k_trace_lane<gr::GenericMetricT<6>" bash scripts/profile_pmc.sh ${T}_dilatonaxion
PROF_CMD="scripts/sibling_workloads.py dual8" PROF_KERNEL="k_trace_lane<gr::GenericMetricT<8>" bash scripts/profile_pmc.sh ${T}_kerrdarkmatter
PROF_CMD="scripts/sibling_workloads.py dual9" PROF_KERNEL="k_trace_lane<gr::GenericMetricT<9>" bash scripts/profile_pmc.sh ${T}_kerrrefractive
PROF_CMD="scripts/sibling_workloads.py dual10" PROF_KERNEL="k_trace_lane<gr::GenericMetricT<10>" bash scripts/profile_pmc.sh ${T}_noz
PROF_CMD="scripts/sibling_workloads.py mesh" PROF_KERNEL="k_trace_lane<gr::KerrFamily<false>, 8>" bash scripts/profile_pmc.sh ${T}_mesh
# drop the bulky raw traces, keep summaries
for d in gpurun_out/prof_${T}_*; do rm -rf $d/trace $d/pmcA $d/pmcB $d/pmcC $d/pmcD $d/pmcE $d/pmcF; done
