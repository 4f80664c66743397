#!/bin/bash
# diag + GPU tests + bench (+ optional rocprof if PROF=1).  Outputs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/gpu_diag.py > gpurun_out/diag.log 2>&1; echo "diag rc=$?"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
if [ "$PROF" = "1" ]; then
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o run -- python "$OLDPWD/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof.log" 2>&1 ); echo "rocprof rc=$?"
fi
echo "==== diag"; grep -E "=====|max_err|FAILED|Error|error" gpurun_out/diag.log | awk '{ if ($0 ~ /max_err/) { if ($3+0 > 1e-4 || $0 ~ /planes|model|decode\./) print } else print }' | head -80
echo "==== timings"; sed -n '/timings/,$p' gpurun_out/diag.log
echo "==== pytest"; tail -n 15 gpurun_out/pytest.log
echo "==== bench"; tail -n 3 gpurun_out/bench.log
