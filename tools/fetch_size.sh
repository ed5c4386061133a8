#!/bin/bash
# HBM-side read / write volume of the kernels matching a name: two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) of one command.
#   tools/fetch_size.sh OUTDIR KERNEL_SUBSTRING python3 tools/xyz.py args...        (counter units: KiB; see DESIGN for the gfx950 factor)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; PAT=$2; shift 2
case $O in /*) ;; *) O=$R/$O;; esac
mkdir -p $O
cd $R
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/p1 -- "$@" > $O/p1.log 2>&1 || { echo "pass 1 failed"; tail -3 $O/p1.log; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p2 -- "$@" > $O/p2.log 2>&1 || { echo "pass 2 failed"; tail -3 $O/p2.log; }
python3 tools/pmc_summary.py $O "$PAT" > $O/summary.json
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
python3 - $O/summary.json <<'PY'
import json, sys
for k, v in json.load(open(sys.argv[1])).items():
    print(k, {c: round(x["per_dispatch"], 1) for c, x in v.items()}, "dispatches", next(iter(v.values()))["dispatches"])
PY
