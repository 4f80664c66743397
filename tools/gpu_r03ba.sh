#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03ba}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2 3 4; do timeout 300 python bench.py --no-extra --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; python - <<PY
import json
d=json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
print("run $i", round(d["ms_per_step"],4), round(d["value"],1), round(d["step_ms_median"],4), round(d["step_ms_max"],4), d["steps"], round(d["roofline"]["frac"],4), d["roofline"]["traffic"])
PY
done
