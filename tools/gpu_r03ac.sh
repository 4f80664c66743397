#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03ac}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
for lib in prev cur; do
  [ $lib = prev ] && export GIGA_DIAG_LIB=$R/giga_amd/lib/diag/libgiga_prev.so || unset GIGA_DIAG_LIB
  echo "== $lib"; GIGA_C4_C2=fp32,fp16x3 GIGA_C4_MODES=default GIGA_C4_PRECS=fp16 GIGA_C4_REPS=2 timeout 600 python tools/gpu_c4_small.py 1 2 8 32 2> $O/c4_$lib.err | tee $O/c4_$lib.txt; tail -n 2 $O/c4_$lib.err
done
