#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04f; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x > $O/tests.log 2>&1; tail -n 12 $O/tests.log
step timeout -k 10 900 python tools/rank_proxy.py 8 256 512 > $O/rank_proxy.json 2> $O/rank_proxy.err; cat $O/rank_proxy.json; tail -n 3 $O/rank_proxy.err
