#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03ad}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_training.py -q -x -n 1 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 $O/pytest.log | cut -c1-400
timeout 600 python tools/gpu_train_ab.py 2 2> $O/train.err | tee $O/train.txt; tail -n 2 $O/train.err
