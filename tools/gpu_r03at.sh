#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03at}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -n 1 -k "several_streams" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 15 $O/pytest.log | cut -c1-300
