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
for w in $WL; do python tools/make_traffic_table.py gpurun_out/traffic/$w gpurun_out/traffic/$w.json $w; done
