#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03o}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
for d in 0 1; do echo "== GIGA_CONV_XCD=$d"; GIGA_CONV_XCD=$d timeout 300 python tools/gpu_stage_all.py 32 8 > $O/stages_xcd$d.txt 2>&1; cat $O/stages_xcd$d.txt | grep -v amdgpu.ids; done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f16_exact.py -q -n 1 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 8 $O/pytest.log | cut -c1-400
