#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04e; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 600 python -m pytest tests/test_mlp.py tests/test_gpu_config4.py -q -m gpu > $O/tests.log 2>&1; tail -n 3 $O/tests.log
step timeout -k 10 600 python tools/mlp_bench.py > $O/mlp_bench.json 2> $O/mlp_bench.err; grep -E "seconds|voxels_per_s" $O/mlp_bench.json
cd /tmp && export TMPDIR=/tmp
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/mlp_bwd_only.py > $O/trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bwd_kernel_stats.csv; head -7 $O/bwd_kernel_stats.csv | cut -c1-120
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
