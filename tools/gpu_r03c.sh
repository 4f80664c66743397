#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03c}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_f16_exact.py -q -x 2>&1 | tail -n 40 | cut -c1-400 > $O/f16_exact.log; cat $O/f16_exact.log
timeout 1500 python -m pytest tests -m gpu -q -n 1 --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 30 $O/pytest.log | cut -c1-300
bash tools/gpu_pmc.sh "c4step c4step_x3" > $O/pmc.txt 2>&1; grep -E "==|decoder|conv16_kernelIDF16|mega" $O/pmc.txt | cut -c1-260
