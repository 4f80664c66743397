#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03u}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 400 python tools/gpu_unet_small.py 1 2 4 8 10 16 32 > $O/unet_small.txt 2> $O/unet_small.err; echo "rc=$?"; cat $O/unet_small.txt; tail -n 5 $O/unet_small.err
GIGA_UNET_GROUPED=1 timeout 400 python tools/gpu_unet_small.py 8 16 32 64 > $O/unet_grouped.txt 2> $O/unet_grouped.err; echo "rc=$?"; cat $O/unet_grouped.txt; tail -n 5 $O/unet_grouped.err
( export GIGA_DIAG_LIB=$R/giga_amd/lib/diag/libgiga_trace.so; GIGA_DIAG_B=1 timeout 200 python tools/gpu_unet_trace.py fp16 > $O/trace_fp16.txt 2>&1; cat $O/trace_fp16.txt )
timeout 1200 python -m pytest tests -m gpu -q -x -n 1 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 6 $O/pytest.log | cut -c1-600
