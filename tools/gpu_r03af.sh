#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; T=${1:-r03af}
O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_training.py -q -x -n 1 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 $O/pytest.log | cut -c1-400
timeout 600 python tools/gpu_train_ab.py 2 2> $O/train.err | tee $O/train.txt; tail -n 2 $O/train.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train16 -o t -- python $R/tools/gpu_prof.py train_bf16_flat 10 > $O/prof_train_bf16.log 2>&1 ); echo "rocprof train bf16 rc=$?"
python tools/prof_summary.py /tmp/prof_train16 $O/train_bf16_flat_kernel_stats.txt; head -8 $O/train_bf16_flat_kernel_stats.txt | cut -c1-160; grep -E "plane_gather|fillBuffer|linear_wgrad|decoder_bwd" $O/train_bf16_flat_kernel_stats.txt | cut -c1-160
