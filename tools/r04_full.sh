#!/bin/bash
# the whole -m gpu suite, then the bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04_full; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed/timed out (rc $rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 1500 python -m pytest tests/ -q -m gpu -x --durations=15 > $O/tests.log 2>&1; tail -n 30 $O/tests.log
step timeout -k 10 900 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err; python3 - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
for c in d.get("cg_mg", []): print("cg_mg", c.get("grid"), c.get("iterations"), c.get("iterations_per_s"), c.get("seconds_samples"))
m=d.get("mlp_forward",{}); print("mlp", m.get("voxels_per_s"), m.get("backward",{}).get("seconds"))
for c in d.get("degree2_cg_mg", []): print("q2 cg", c.get("grid"), c.get("iterations_per_s"))
for c in d.get("degree2_spmv", []): print("q2 spmv", c.get("grid"), c.get("frac_of_8TBs"))
PY
