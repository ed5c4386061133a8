#!/bin/bash
# round 4, first GPU call: new assertions, rank proxy, counters of the default MLP kernel and of the current degree-2 marching apply
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04a; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "parameterisations or headline" > $O/tests1.log 2>&1; tail -3 $O/tests1.log
step timeout -k 10 600 python -m pytest tests/test_gpu_degree2.py -q -m gpu -k "config5_size" > $O/tests2.log 2>&1; tail -3 $O/tests2.log
step timeout -k 10 900 python tools/rank_proxy.py 8 256 512 > $O/rank_proxy.json 2> $O/rank_proxy.err; tail -40 $O/rank_proxy.json; tail -3 $O/rank_proxy.err
cd /tmp && export TMPDIR=/tmp
# MLP default kernel: SQ + TCC counters (separate passes)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  step timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/mlp/p$i -- python3 $R/tools/mlp_fwd_only.py > $O/mlp_p$i.log 2>&1 || true
  tail -1 $O/mlp_p$i.log
done
cd $R && python3 tools/pmc_summary.py $O/mlp k_mlp_forward_x3 > $O/mlp_x3_pmc.json; cat $O/mlp_x3_pmc.json | head -80
step timeout -k 10 600 bash tools/fetch_size.sh $O/q2 k_apply_q2_march python3 tools/q2_only.py 512
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete
