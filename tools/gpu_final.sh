#!/bin/bash
# End-of-round evidence in ONE GPU call: GPU tests, smoke, the default bench line, rocprofv3 kernel stats of the bench
# command and of a training step, HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE separately).  Outputs: gpurun_out/final/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export GIGA_COMMIT=${GIGA_COMMIT:-unknown}
O=$R/gpurun_out/final; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 2 $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_bench.log 2>&1 ); echo "rocprof bench rc=$?"
python tools/prof_summary.py /tmp/prof_bench $O/bench_kernel_stats.txt
# the same command without the extra legs: the dominant launch's rocprofv3 average in the SAME context as bench.py's HIP events
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_core -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/prof_bench_core.log 2>&1 ); echo "rocprof bench core rc=$?"
python tools/prof_summary.py /tmp/prof_core $O/bench_core_kernel_stats.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o t -- python $R/tools/gpu_prof.py train_flat 10 > $O/prof_train.log 2>&1 ); echo "rocprof train rc=$?"
python tools/prof_summary.py /tmp/prof_train $O/train_kernel_stats.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train16 -o t -- python $R/tools/gpu_prof.py train_bf16_flat 10 > $O/prof_train_bf16.log 2>&1 ); echo "rocprof train bf16 rc=$?"
python tools/prof_summary.py /tmp/prof_train16 $O/train_bf16_kernel_stats.txt
bash tools/gpu_traffic.sh "c2 c4step c4step_x3" > $O/traffic.txt 2>&1; echo "traffic rc=$?"; cp gpurun_out/traffic/*.json $O/ 2>/dev/null
bash tools/gpu_pmc.sh "c4step c4step_x3 c2" > $O/pmc.txt 2>&1; echo "pmc rc=$?"; grep -E "==|giga" $O/pmc.txt | cut -c1-250 | tail -n 45
# the conv32 U-Net kernels under the same counters at 32 scenes, where conv16 is the default (GIGA_CONV32=1 forces them for the whole process)
rm -rf gpurun_out/pmc; GIGA_CONV32=1 bash tools/gpu_pmc.sh "c4step c4step_x3" > $O/pmc_conv32.txt 2>&1; echo "pmc conv32 rc=$?"; grep -E "==|unet" $O/pmc_conv32.txt | cut -c1-250
# conv16 against conv32 in one process, interleaved (the A/B that survives box drift), and the sustained-load clocks
GIGA_PRECS=fp16,fp16x3 timeout 600 python tools/gpu_unet_ab.py 1 2 8 32 128 > $O/unet_ab.txt 2>&1; echo "unet_ab rc=$?"; grep B= $O/unet_ab.txt | cut -c1-180
timeout 300 python tools/gpu_sustained_power.py 32 128 > $O/sustained_power.txt 2>&1; grep B= $O/sustained_power.txt | cut -c1-200
tail -n 45 $O/traffic.txt
