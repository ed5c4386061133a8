#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04h; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "apply" > $O/tests.log 2>&1; tail -n 5 $O/tests.log
step timeout -k 10 600 python tools/apply_lx.py 512 256 > $O/apply_lx.txt 2>&1; cat $O/apply_lx.txt
