#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03n}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 python tools/gpu_stage_all.py 32 1 > $O/stages.txt 2>&1; cat $O/stages.txt | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c4_shapes.py -q -n 1 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 12 $O/pytest.log | cut -c1-400
