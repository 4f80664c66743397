#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03ae}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train16 -o t -- python $R/tools/gpu_prof.py train_bf16_flat 10 > $O/prof_train_bf16.log 2>&1 ); echo "rocprof train bf16 rc=$?"
python tools/prof_summary.py /tmp/prof_train16 $O/train_bf16_flat_kernel_stats.txt; head -14 $O/train_bf16_flat_kernel_stats.txt | cut -c1-160; grep -E "plane_scatter|fillBuffer|linear_wgrad" $O/train_bf16_flat_kernel_stats.txt | cut -c1-160
