#!/bin/bash
# fragment-order weights: MLP parity tests + rates
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04b; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 600 python -m pytest tests/test_mlp.py tests/test_gpu_config4.py -q -m gpu -x > $O/tests.log 2>&1; tail -n 3 $O/tests.log
step timeout -k 10 600 python tools/mlp_bench.py > $O/mlp_bench.json 2> $O/mlp_bench.err; cat $O/mlp_bench.json
step timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "parameterisations" > $O/tests_pcg.log 2>&1; tail -n 2 $O/tests_pcg.log
