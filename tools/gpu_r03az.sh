#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03az}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do timeout 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_$i.log 2>&1; echo "pytest run $i rc=$?"; tail -n 2 $O/pytest_$i.log | cut -c1-200; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log
