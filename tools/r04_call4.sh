#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04d; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 600 python -m pytest tests/test_mlp.py tests/test_gpu_config4.py -q -m gpu > $O/tests.log 2>&1; tail -n 6 $O/tests.log
step timeout -k 10 600 python tools/mlp_bwd_probe.py > $O/bwd_probe.log 2>&1; grep -c NaN $O/bwd_probe.log; tail -n 2 $O/bwd_probe.log
cd /tmp && export TMPDIR=/tmp
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/mlp_bwd_only.py > $O/trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bwd_kernel_stats.csv; head -9 $O/bwd_kernel_stats.csv | cut -c1-150
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  step timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc/p$i -- python3 $R/tools/mlp_bwd_only.py > $O/pmc_p$i.log 2>&1 || true
done
cd $R && python3 tools/pmc_summary.py $O/pmc k_mlp > $O/mlp_bwd_pmc.json
python3 - $O/mlp_bwd_pmc.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    print(k)
    print("  "+"  ".join("%s=%.3g"%(c.replace("SQ_","").replace("_sum",""),x['per_dispatch']) for c,x in v.items()))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete
