#!/bin/bash
# round 4: conv32 with LDS-resident weights, 8 waves: tests, A/B, traces
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4c; mkdir -p $O; export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_f16_exact.py tests/test_gpu_parity.py tests/test_gpu_c4_shapes.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
GIGA_CONV32=1 GIGA_PRECS=fp16,fp16x3 timeout 300 python tools/gpu_unet_small.py 1 8 32 128 > $O/unet_small_conv32_1.log 2>&1; echo "unet_small rc=$?"
GIGA_CONV32=0 GIGA_PRECS=fp16x3 timeout 300 python tools/gpu_unet_small.py 1 32 > $O/unet_small_conv32_0.log 2>&1; echo "unet_small rc=$?"
for B in 32 1; do for P in fp16 fp16x3; do
  GIGA_DIAG_B=$B GIGA_DIAG_LIB=$PWD/giga_amd/lib/diag/libgiga_trace.so timeout 120 python tools/gpu_c32_trace.py $P > $O/c32_trace_${P}_B$B.log 2>&1; echo "trace $P B=$B rc=$?"
done; done
cat $O/unet_small_conv32_1.log $O/unet_small_conv32_0.log $O/c32_trace_fp16_B32.log $O/c32_trace_fp16_B1.log $O/c32_trace_fp16x3_B32.log
