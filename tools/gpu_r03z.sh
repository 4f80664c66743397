#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03z}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 python tools/gpu_train_ops.py > $O/train_ops.txt 2>&1; echo rc=$?; grep -v "^-" $O/train_ops.txt | grep -i -B0 -A7 "copy\|fill\|Memcpy\|Memset\|label" | cut -c1-330 | head -120
