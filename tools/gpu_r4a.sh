#!/bin/bash
# round 4, first conv32 call: GPU tests, then encoder A/B (conv16 vs conv32) in f16
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4a; mkdir -p $O; export PYTHONPATH=.
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for c in 1 0; do
  GIGA_CONV32=$c GIGA_PRECS=fp16 timeout 300 python tools/gpu_unet_small.py 1 8 32 128 > $O/unet_small_conv32_$c.log 2>&1; echo "unet_small conv32=$c rc=$?"
  GIGA_CONV32=$c GIGA_PRECS=fp16 timeout 300 python tools/gpu_stage_all.py 32 1 > $O/stage_all_conv32_$c.log 2>&1; echo "stage_all conv32=$c rc=$?"
done
cat $O/unet_small_conv32_*.log $O/stage_all_conv32_1.log
