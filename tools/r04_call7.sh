#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04g; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 600 python tools/rank_proxy.py levels 8 512 > $O/levels512.jsonl 2> $O/levels512.err; cat $O/levels512.jsonl | cut -c1-260
step timeout -k 10 600 python tools/rank_proxy.py levels 8 256 > $O/levels256.jsonl 2> $O/levels256.err; cat $O/levels256.jsonl | cut -c1-260
cd /tmp && export TMPDIR=/tmp
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t512 -- python3 $R/tools/rank_proxy.py one 8 512 > $O/t512.log 2>&1
find $O/t512 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/proxy512_kernel_stats.csv; head -30 $O/proxy512_kernel_stats.csv | cut -c1-160
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t256 -- python3 $R/tools/rank_proxy.py one 8 256 > $O/t256.log 2>&1
find $O/t256 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/proxy256_kernel_stats.csv; head -30 $O/proxy256_kernel_stats.csv | cut -c1-160
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
