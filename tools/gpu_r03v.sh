#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03v}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
export GIGA_DIAG_LIB=$R/giga_amd/lib/diag/libgiga_trace.so
GIGA_DIAG_B=1 timeout 200 python tools/gpu_unet_trace.py fp16 > $O/trace_fp16.txt 2>&1; cat $O/trace_fp16.txt
GIGA_DIAG_B=1 GIGA_DIAG_PREC=fp16 timeout 200 python tools/gpu_conv_trace.py 0 3 5 7 > $O/conv_trace_fp16.txt 2>&1; cat $O/conv_trace_fp16.txt
