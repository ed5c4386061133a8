#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04i; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 600 python -m pytest tests/test_gpu_gs_march.py -q -m gpu -x > $O/tests.log 2>&1; tail -n 5 $O/tests.log
for lib in "" ndr_amd/csrc/ab_libs/libvfem_gsm_r6.so ndr_amd/csrc/ab_libs/libvfem_r04_before_gsm.so ""; do
  VFEM_LIB=${lib:+$R/$lib} step timeout -k 10 300 python tools/gs_sweep_time.py 512 256 >> $O/sweep_times.txt 2>&1
done
cat $O/sweep_times.txt | grep -v amdgpu.ids
