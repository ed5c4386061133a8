#!/bin/bash
# End-of-round measurement set (run on the GPU box from the repo root):  bash tools/final_profiles.sh r03
# writes gpurun_out/<tag>_* ; the summaries to be judged are copied to profiles/ by hand afterwards
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== bench"; python3 bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err; tail -c 600 $O/${TAG}_bench_n1.json
echo "== bench under rocprofv3 --kernel-trace --stats"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_bench_prof -- python3 $R/bench.py > $O/${TAG}_bench_profiled.json 2> $O/${TAG}_bench_profiled.err )
f=$(find $O/${TAG}_bench_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_bench_kernel_stats.csv
echo "== apply traffic (PMC)"; bash tools/fetch_size.sh $O/${TAG}_apply_pmc k_apply_dma python3 tools/apply_only.py 512 6
echo "== marching sweep traffic (PMC)"; bash tools/fetch_size.sh $O/${TAG}_gsm_pmc k_gs_march python3 tools/gs_march_only.py 512 2
echo "== level-1 sweep traffic (PMC)"; bash tools/fetch_size.sh $O/${TAG}_l1m_pmc k_l1_merged python3 tools/l1_sweep_only.py 512 1 4
for n in 512 256; do
  lv=6; [ $n = 256 ] && lv=5
  echo "== pcg $n"
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_pcg${n}_prof -- python3 $R/tools/pcg_only.py $n $lv > $O/${TAG}_pcg${n}.log 2>&1 )
  f=$(find $O/${TAG}_pcg${n}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_pcg${n}_kernel_stats.csv
  grep iterations_per $O/${TAG}_pcg${n}.log
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
echo done
