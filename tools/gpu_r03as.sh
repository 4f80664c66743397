#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03as}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
GIGA_C4_MODES=default GIGA_C4_PRECS=fp16 GIGA_C4_REPS=2 timeout 600 python tools/gpu_c4_small.py 1 8 32 128 2> $O/c4.err | tee $O/c4.txt; tail -n 2 $O/c4.err
timeout 600 python -m pytest tests/test_gpu_f16_exact.py tests/test_gpu_parity.py -q -x -n 1 -k "f16 or persistent or lattice" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log | cut -c1-300
