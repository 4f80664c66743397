#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4f; mkdir -p $O; export PYTHONPATH=.
timeout 300 python -m pytest tests/test_gpu_f16_exact.py tests/test_gpu_c4_shapes.py tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
GIGA_PRECS=fp16,fp16x3 timeout 300 python tools/gpu_unet_small.py 1 8 32 128 > $O/unet_small.log 2>&1; echo "unet_small rc=$?"
for B in 32; do for P in fp16; do
  GIGA_DIAG_B=$B GIGA_DIAG_LIB=$PWD/giga_amd/lib/diag/libgiga_trace.so timeout 120 python tools/gpu_c32_trace.py $P > $O/c32_trace_${P}_B$B.log 2>&1; echo "trace $P B=$B rc=$?"
done; done
cat $O/unet_small.log $O/c32_trace_fp16_B32.log
