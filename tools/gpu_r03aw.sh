#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03aw}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_training.py -q -x -n 1 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log | cut -c1-300
GIGA_TRAIN_PRECS=bf16 timeout 600 python tools/gpu_train_ab.py 2 2> $O/train.err | tee $O/train.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train16 -o t -- python $R/tools/gpu_prof.py train_bf16_flat 10 > $O/prof_train_bf16.log 2>&1 ); echo "rocprof rc=$?"
python tools/prof_summary.py /tmp/prof_train16 $O/train_bf16_flat_kernel_stats.txt; grep -E "wgrad_bf16|wgrad3_reduce" $O/train_bf16_flat_kernel_stats.txt | cut -c1-150
