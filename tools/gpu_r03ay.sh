#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03ay}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
GIGA_C4_C2=fp32,fp16x3 GIGA_C4_MODES=default,layers GIGA_C4_PRECS=fp16 GIGA_C4_REPS=2 timeout 600 python tools/gpu_c4_small.py 1 8 32 2> $O/c4.err | tee $O/c4.txt; tail -n 2 $O/c4.err
