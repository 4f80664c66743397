#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03av}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
( export GIGA_DIAG_LIB=$R/giga_amd/lib/diag/libgiga_trace.so; GIGA_DIAG_TRAIN_PREC=bf16 timeout 200 python tools/gpu_wgrad_trace.py > $O/wgrad_trace_bf16.txt 2>&1; grep -E "ev ?[0-8] |ev30|ev31" $O/wgrad_trace_bf16.txt )
timeout 900 python -m pytest tests/test_gpu_training.py -q -x -n 1 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log | cut -c1-300
GIGA_TRAIN_PRECS=bf16 timeout 600 python tools/gpu_train_ab.py 2 2> $O/train.err | tee $O/train.txt
