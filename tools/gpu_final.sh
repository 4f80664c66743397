#!/bin/bash
# End-of-round evidence in ONE GPU call: GPU tests, smoke, the default bench line, rocprofv3 kernel stats of the bench
# command and of a training step, HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE separately).  Outputs: gpurun_out/final/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export GIGA_COMMIT=${GIGA_COMMIT:-unknown}
O=$R/gpurun_out/final; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 2 $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log
GIGA_BENCH_EXTRA=$O/bench_extra.json timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
export GIGA_BENCH_EXTRA=/tmp/bench_extra_prof.json      # (the profiled reruns below must not overwrite the side file of the run above)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_bench.log 2>&1 ); echo "rocprof bench rc=$?"
python tools/prof_summary.py /tmp/prof_bench $O/bench_kernel_stats.txt
# the same command without the extra legs: the dominant launch's rocprofv3 average in the SAME context as bench.py's HIP events
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_core -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/prof_bench_core.log 2>&1 ); echo "rocprof bench core rc=$?"
python tools/prof_summary.py /tmp/prof_core $O/bench_core_kernel_stats.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o t -- python $R/tools/gpu_prof.py train_flat 10 > $O/prof_train.log 2>&1 ); echo "rocprof train rc=$?"
python tools/prof_summary.py /tmp/prof_train $O/train_kernel_stats.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train16 -o t -- python $R/tools/gpu_prof.py train_bf16_flat 10 > $O/prof_train_bf16.log 2>&1 ); echo "rocprof train bf16 rc=$?"
python tools/prof_summary.py /tmp/prof_train16 $O/train_bf16_kernel_stats.txt
bash tools/gpu_traffic.sh "c2 c4step c4step_x3 train_flat train_bf16_flat" > $O/traffic.txt 2>&1; echo "traffic rc=$?"; cp gpurun_out/traffic/*.json $O/ 2>/dev/null
bash tools/gpu_pmc.sh "c4step c4step_x3 c2 train_bf16_flat" > $O/pmc.txt 2>&1; echo "pmc rc=$?"; grep -E "==|giga" $O/pmc.txt | cut -c1-250 | tail -n 70
# round 5: every launch of a c5 step in order (event brackets on the launch's own stream), both precisions
timeout 300 python tools/gpu_c5_launches.py bf16 > $O/c5_bf16_launches.txt 2>&1; timeout 300 python tools/gpu_c5_launches.py fp32 > $O/c5_fp32_launches.txt 2>&1
timeout 300 python tools/gpu_c5_time.py > $O/c5_step_times.txt 2>&1; tail -n 3 $O/c5_step_times.txt
tail -n 45 $O/traffic.txt
