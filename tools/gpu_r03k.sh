#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03k}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
DEC_LAT_MODES=0,1,1n16 timeout 600 python tools/gpu_dec_lat.py 8,32,128 > $O/dec_lat.txt 2> $O/dec_lat.err; echo "dec_lat rc=$?"; python - <<PY
import json
for l in open("$O/dec_lat.txt"):
    d=json.loads(l); print(d["prec"],d["scenes"],{k:v for k,v in d.items() if k.startswith(("ms_","alg_","err_1n"))})
PY
tail -n 3 $O/dec_lat.err
for p in fp16; do GIGA_LAT_NW=12 GIGA_DIAG_LIB=$R/giga_amd/lib/diag/libgiga_trace.so GIGA_DIAG_B=32 timeout 150 python tools/gpu_dec_trace.py $p > $O/trace_$p.txt 2>&1; tail -n 13 $O/trace_$p.txt; done
