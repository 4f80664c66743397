#!/bin/bash
# HBM traffic of every kernel of a workload of tools/gpu_prof.py (default: c2 and c4step): FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes (TCC slots), kernel-trace + pmc only.  Prints per-kernel averages (FETCH_SIZE doubled per the gfx950 note)
# and writes profiles-ready JSON tables gpurun_out/traffic/<workload>.json (stamped with the commit in $GIGA_COMMIT).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
mkdir -p gpurun_out/traffic; export TMPDIR=/tmp
WL=${1:-"c2 c4step"}
cd /tmp
for w in $WL; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/traffic/$w/$c -o t --output-format csv -- python $R/tools/gpu_prof.py $w 3 > $R/gpurun_out/traffic/$w.$c.log 2>&1
    echo "pass $w $c rc=$?"
  done
done
cd $R
for w in $WL; do
  case $w in train*) python tools/make_traffic_table.py gpurun_out/traffic/$w gpurun_out/traffic/$w.json $w 4;;   # 3 iterations + 1 warm-up: per-step totals
             *) python tools/make_traffic_table.py gpurun_out/traffic/$w gpurun_out/traffic/$w.json $w;; esac
done
# c5: one file for both arithmetic modes of the training step (bench.py reads profiles/r0N_traffic_c5.json)
if [ -f gpurun_out/traffic/train_flat.json ] && [ -f gpurun_out/traffic/train_bf16_flat.json ]; then
  python - <<'PY'
import json
t = {"_about": "HBM traffic of the c5 training step (B = 32 scenes, 1 grasp + 2048 occupancy queries, FlatAdam), every kernel of the step: "
               "two rocprofv3 PMC passes per mode (FETCH_SIZE doubled per the gfx950 note, WRITE_SIZE), tools/gpu_traffic.sh; "
               "bytes = per launch, bytes_per_step = bytes x launches per step",
     "fp32": json.load(open("gpurun_out/traffic/train_flat.json")), "bf16": json.load(open("gpurun_out/traffic/train_bf16_flat.json"))}
json.dump(t, open("gpurun_out/traffic/c5.json", "w"), indent=1)
print("c5 bytes per step: fp32 %.1f MB, bf16 %.1f MB" % (t["fp32"]["bytes_per_step"] / 1e6, t["bf16"]["bytes_per_step"] / 1e6))
PY
fi
