#!/bin/bash
# round 3, GPU job 12: long soak of the final build; default bench line with the fp64_issue field; reference benchmark suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench.err; tail -c 1500 $O/bench_default.json | head -c 700; echo
timeout 2400 python3 scripts/soak.py 20000 2027 > $O/soak_20000_seed2027.log 2>&1; head -1 $O/soak_20000_seed2027.log | cut -c1-500
timeout 900 python3 scripts/soak32.py 2600 32 > $O/soak32_2600.log 2>&1; tail -3 $O/soak32_2600.log | cut -c1-300
timeout 600 python3 scripts/benchmark_tracing.py > $O/benchmark_tracing.log 2>&1; tail -12 $O/benchmark_tracing.log
