#!/bin/bash
# GPU tests + smoke + default bench line in one call.  Output: gpurun_out/$1/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03a}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 25 $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 3 $O/smoke.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -n 5 $O/bench.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench.json"))
    print({k:d[k] for k in ("value","ms_per_step","launches_per_step")}, d["roofline"]["worst_stage"], d["roofline"]["argmax_stage"])
    e=d.get("extra",{})
    for k,v in e.items():
        if isinstance(v,dict): print(k,{kk:v[kk] for kk in v if kk in("ms_per_step","step_ms_median","scenes_per_sec","checked_vs_oracle")})
        elif isinstance(v,list): print(k,[(s["scenes"],round(s["ms_per_step"],4),round(s["decoder_ms"],4),round(s["decoder_frac_of_f16_mfma_peak"],3)) for s in v])
        else: print(k,v)
except Exception as ex: print("bench parse failed", ex)
PY
