#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03bc}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 python tools/gpu_c2_host.py > $O/host.txt 2>&1; head -60 $O/host.txt | cut -c1-160
