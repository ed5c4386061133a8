#!/bin/bash
# rocprofv3 passes for one command: kernel trace + stats, then the SQ / TCC counter passes (each in a run of its own).
#   tools/prof_pmc.sh OUTDIR KERNEL_SUBSTRING python3 tools/xyz.py args...
# writes OUTDIR/summary.json (per-kernel counter sums, tools/pmc_summary.py) and OUTDIR/kernel_stats.csv
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; PAT=$2; shift 2
case $O in /*) ;; *) O=$R/$O;; esac
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- "$@" > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $O/p$i -- "$@" > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; }
done
python3 tools/pmc_summary.py $O "$PAT" > $O/summary.json
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
head -8 $O/kernel_stats.csv
