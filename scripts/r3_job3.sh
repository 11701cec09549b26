#!/bin/bash
# round 3, GPU job 3: full GPU suite; A/B of the cold lane storage (Kerr: on / resident rotation constants / off;
# Johannsen: 2 waves / 3 waves / 2 waves without the cold store)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs > $O/pytest.log 2>&1; tail -25 $O/pytest.log
timeout 600 python3 scripts/ab_bench.py abv/k0.so abv/k1.so abv/k2.so --rounds 10 > $O/ab_kerr.log 2>&1; cat $O/ab_kerr.log
timeout 600 python3 scripts/ab_bench.py abv/j2.so abv/j3.so abv/j2nc.so --rounds 10 --size 1024 --workload johannsen > $O/ab_johannsen.log 2>&1; cat $O/ab_johannsen.log
