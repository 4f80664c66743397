#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03au}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
export GIGA_DIAG_LIB=$R/giga_amd/lib/diag/libgiga_trace.so
GIGA_DIAG_TRAIN_PREC=bf16 timeout 200 python tools/gpu_wgrad_trace.py > $O/wgrad_trace_bf16.txt 2>&1; cat $O/wgrad_trace_bf16.txt | tail -30
