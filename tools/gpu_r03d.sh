#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03d}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 600 python tools/gpu_dec_lat.py 8,32,128 > $O/dec_lat.txt 2> $O/dec_lat.err; echo "dec_lat rc=$?"; cat $O/dec_lat.txt | cut -c1-330; tail -n 3 $O/dec_lat.err
timeout 1500 python -m pytest tests -m gpu -q -n 1 --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 30 $O/pytest.log | cut -c1-600
bash tools/gpu_pmc.sh "c4step c4step_x3" > $O/pmc.txt 2>&1; grep -E "==|decoder" $O/pmc.txt | cut -c1-260
