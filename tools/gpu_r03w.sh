#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03w}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 1200 python -m pytest tests -m gpu -q -x -n 1 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 $O/pytest.log | cut -c1-600
timeout 400 python tools/gpu_unet_small.py 1 3 10 32 > $O/unet_small.txt 2> $O/unet_small.err; echo "rc=$?"; cat $O/unet_small.txt; tail -n 3 $O/unet_small.err
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -n 3 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","launches_per_step") if k in d})
e=d.get("extra",{})
for k,v in e.items():
    if isinstance(v,dict) and "ms_per_step" in v: print(k, v["ms_per_step"], v.get("decoder_ms"), v.get("checked_vs_oracle"))
    elif isinstance(v,list): 
        for r in v: print(k, {kk:r[kk] for kk in r if kk in ("scenes","ms_per_step","decoder_ms","decoder_frac","precision")})
PY
