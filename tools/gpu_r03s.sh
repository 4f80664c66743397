#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03s}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 400 python tools/gpu_unet_small.py 1 2 4 8 10 > $O/unet_small.txt 2> $O/unet_small.err; echo "rc=$?"; cat $O/unet_small.txt; tail -n 5 $O/unet_small.err
