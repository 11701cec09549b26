#!/bin/bash
# round 3, GPU job 6: A/B of the weighted-sum / rotation-constant / late-parameter changes on the Kerr kernels; full GPU
# suite on the new build; end-point call into pinned memory with bands on two streams
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 600 python3 scripts/ab_bench.py abv/h0.so abv/n1.so abv/n2.so abv/n3.so --rounds 12 > $O/ab_kerr.log 2>&1; cat $O/ab_kerr.log
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs > $O/pytest.log 2>&1; tail -12 $O/pytest.log
timeout 600 python3 scripts/endpoints_pinned_time.py 2048 > $O/endpoints_pinned.log 2>&1; head -6 $O/endpoints_pinned.log
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench.err; tail -c 700 $O/bench_default.json
