#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04j; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 1500 python -m pytest tests/test_gpu_distributed_q2.py tests/test_gpu_full_size.py tests/test_gpu_generic.py tests/test_gpu_gs_march.py tests/test_gpu_parity.py tests/test_mlp.py tests/test_train_loop.py tests/test_io_formats.py -q -m gpu > $O/tests.log 2>&1; tail -n 8 $O/tests.log
