#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3w; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "pinned or block or endpoint" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
