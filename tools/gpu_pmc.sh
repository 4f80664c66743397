#!/bin/bash
# PMC pass (kernel-trace + counters only) per workload; prints per-kernel averages and MFMA-pipe utilisation.
# usage: bash tools/gpu_pmc.sh "c4dec enc32 ..."
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
WL=${1:-"c4dec enc32"}
cd /tmp
for w in $WL; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
     -d $R/gpurun_out/pmc/$w -o $w --output-format csv -- python $R/tools/gpu_prof.py $w 3 > $R/gpurun_out/pmc/$w.log 2>&1
  echo "pmc $w rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]][0] += float(r["Counter_Value"]); agg[k][r["Counter_Name"]][1] += 1
    print("==", f)
    for kn, cs in agg.items():
        if not ("giga" in kn): continue
        g = {c: v[0] / v[1] for c, v in cs.items()}
        gui = g.get("GRBM_GUI_ACTIVE", 0) / 8.0           # summed over 8 XCDs
        mfma = g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        wc = g.get("SQ_WAVE_CYCLES", 1)
        util = mfma / (gui * 1024) if gui else 0
        print(f"  {kn[:92]:92s} gpu_cycles {gui:10.0f} mfma_util {util:5.2f} wait_any {g.get('SQ_WAIT_ANY',0)/wc:4.2f} "
              f"wait_inst {g.get('SQ_WAIT_INST_ANY',0)/wc:4.2f} active {g.get('SQ_ACTIVE_INST_ANY',0)/wc:4.2f} "
              f"valu/mfma {g.get('SQ_INSTS_VALU',0)/max(g.get('SQ_INSTS_MFMA',1),1):5.1f} bankconf {g.get('SQ_LDS_BANK_CONFLICT',0):.0f}")
PY
