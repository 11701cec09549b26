#!/bin/bash
# profile_pmc.sh for a GR_METRIC_TABULATED workload plus the counters that say where ITS time goes: the scalar data cache
# (coefficient fetches) and LDS.   scripts/profile_tab.sh <tag> [tabkerr|tabc4] [reps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; W=${2:-tabkerr}; REPS=${3:-5}
export PROF_CMD="scripts/sibling_workloads.py $W $REPS" PROF_KERNEL=k_trace
bash scripts/profile_pmc.sh $TAG
OUT=$R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmcE -- python3 $PROF_CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM --output-format csv -d $OUT/pmcF -- python3 $PROF_CMD > /dev/null 2>&1
rocprofv3 --list-avail 2>/dev/null | grep -i -E "^\s*(Name|Counter).*(SQC_|LDS|SMEM|SCA)" | head -120 > $OUT/avail_counters.txt
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for d in ("pmcE", "pmcF"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "k_trace" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        # per launch: rows repeat per dispatch; drop the first two launches
        for k, v in acc.items():
            res[k] = sum(v[2:]) / max(len(v[2:]), 1) if len(v) > 2 else sum(v) / max(len(v), 1)
json.dump(res, open(f"{out}/extra_counters.json", "w"), indent=1)
print(json.dumps(res))
PY
