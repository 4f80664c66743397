#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03t}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
export GIGA_DIAG_LIB=$R/giga_amd/lib/diag/libgiga_trace.so
for p in fp16 fp32; do GIGA_DIAG_B=1 timeout 200 python tools/gpu_unet_trace.py $p > $O/trace_$p.txt 2> $O/trace_$p.err; echo "rc=$?"; cat $O/trace_$p.txt; done
