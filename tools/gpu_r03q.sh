#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03q}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
DEC_LAT_MODES=0,1,1n16 timeout 600 python tools/gpu_dec_lat.py 4,8,32,128 > $O/dec_lat.txt 2> $O/dec_lat.err; echo "dec_lat rc=$?"; python - <<PY
import json
for l in open("$O/dec_lat.txt"):
    d=json.loads(l); print(d["prec"],d["scenes"],{k:v for k,v in d.items() if k.startswith(("ms_","alg_","err_1n"))})
PY
tail -n 3 $O/dec_lat.err
timeout 900 python -m pytest tests/test_gpu_f16_exact.py tests/test_gpu_c4_shapes.py -q -n 1 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 6 $O/pytest.log | cut -c1-600
