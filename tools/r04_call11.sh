#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04k; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests/test_mlp.py tests/test_gpu_distributed_q2.py tests/test_gpu_distributed.py tests/test_gpu_dense.py -q -m gpu > $O/tests.log 2>&1; tail -n 6 $O/tests.log
step timeout -k 10 300 python tools/mlp_bench.py > $O/mlp_bench.json 2>/dev/null; grep -E "voxels_per_s|seconds" $O/mlp_bench.json
