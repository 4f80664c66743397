#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03l}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
DEC_LAT_MODES=0,1n16 timeout 600 python tools/gpu_dec_lat.py 1,2,3,4 > $O/dec_lat.txt 2> $O/dec_lat.err; echo "dec_lat rc=$?"; python - <<PY
import json
for l in open("$O/dec_lat.txt"):
    d=json.loads(l); print(d["prec"],d["scenes"],{k:v for k,v in d.items() if k.startswith(("ms_","alg_","err_"))})
PY
tail -n 3 $O/dec_lat.err
timeout 1500 python -m pytest tests -m gpu -q -n 1 --durations=6 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 16 $O/pytest.log | cut -c1-400
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -n 3 $O/bench.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench.json"))
    print({k:d[k] for k in ("value","ms_per_step","launches_per_step")}, d["roofline"]["worst_stage"], d["roofline"]["argmax_stage"])
    e=d.get("extra",{})
    for k,v in e.items():
        if isinstance(v,dict): print(k,{kk:v[kk] for kk in v if kk in("ms_per_step","step_ms_median","scenes_per_sec","checked_vs_oracle")}, (v.get("roofline") or {}).get("frac"))
        elif isinstance(v,list): print(k,[(s["scenes"],round(s["ms_per_step"],4),round(s["decoder_ms"],4),round(s["decoder_frac_of_f16_mfma_peak"],3)) for s in v])
        else: print(k,v)
except Exception as ex: print("bench parse failed", ex)
PY
