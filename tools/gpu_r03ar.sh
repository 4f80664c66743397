#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03ar}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 600 python tools/gpu_c4_graph.py 1 8 2> $O/graph.err | tee $O/graph.txt; tail -n 3 $O/graph.err
