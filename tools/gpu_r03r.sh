#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03r}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 python tools/gpu_stage_all.py 32 1 > $O/stages.txt 2> $O/stages.err; echo "stages rc=$?"; cat $O/stages.txt
timeout 900 python -m pytest tests/test_gpu_f16_exact.py tests/test_gpu_training.py tests/test_gpu_parity.py -q -n 1 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 6 $O/pytest.log | cut -c1-600
