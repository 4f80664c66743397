#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03ab}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
GIGA_C4_MODES=layers,default GIGA_C4_PRECS=fp16 GIGA_C4_REPS=3 timeout 600 python tools/gpu_c4_small.py 1 8 32 > $O/c4_small.txt 2> $O/c4_small.err; echo rc=$?; cat $O/c4_small.txt; tail -n 3 $O/c4_small.err
