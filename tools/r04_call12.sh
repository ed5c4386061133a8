#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04l; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests/test_mlp.py tests/test_gpu_config4.py tests/test_train_loop.py -q -m gpu > $O/tests.log 2>&1; tail -n 6 $O/tests.log
step timeout -k 10 600 python tools/mlp_bwd_probe.py > $O/bwd_probe.log 2>&1; tail -n 5 $O/bwd_probe.log
