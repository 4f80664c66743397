#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03b}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 600 python tools/gpu_dec_lat.py > $O/dec_lat.txt 2> $O/dec_lat.err; echo "dec_lat rc=$?"; cat $O/dec_lat.txt | cut -c1-600; tail -n 5 $O/dec_lat.err
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 40 $O/pytest.log | cut -c1-300
