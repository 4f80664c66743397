#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03bb}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
for k in 20 50; do GIGA_BENCH_DUMP_STEPS=1 timeout 300 python bench.py --steps $k --warmup 5 --no-extra --no-cpu-baseline > $O/bench_$k.json 2> $O/bench_$k.err; grep step_ms $O/bench_$k.err | cut -c1-700; python -c "
import json; d=json.loads(open('$O/bench_$k.json').read().strip().splitlines()[-1]); print('K=$k', round(d['ms_per_step'],4), round(d['step_ms_median'],4))"; done
