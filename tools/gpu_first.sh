#!/bin/bash
# One GPU call: diagnosis, GPU tests, smoke, bench, rocprof kernel trace.  Outputs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
timeout 900 python tools/gpu_diag.py > gpurun_out/diag.log 2>&1; echo "diag rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r01 -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof.log" 2>&1 ); echo "rocprof rc=$?"
echo "==== diag"; tail -n 120 gpurun_out/diag.log
echo "==== pytest"; tail -n 30 gpurun_out/pytest.log
echo "==== smoke"; tail -n 12 gpurun_out/smoke.log
echo "==== bench"; tail -n 5 gpurun_out/bench.log
