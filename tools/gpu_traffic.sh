#!/bin/bash
# HBM traffic of every kernel of the c2 step: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots),
# kernel-trace + pmc only.  Prints per-kernel averages in bytes (FETCH_SIZE doubled per the gfx950 note).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
mkdir -p gpurun_out/traffic; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/traffic/$c -o c2 --output-format csv -- python $R/tools/gpu_prof.py c2 3 > $R/gpurun_out/traffic/$c.log 2>&1
  echo "pass $c rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/traffic/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t = tot[r["Kernel_Name"]][r["Counter_Name"]]
        t[0] += float(r["Counter_Value"]); t[1] += 1
print(f"{'kernel':90s} {'FETCH_SIZE(KiB)':>16s} {'x2 bytes':>14s} {'WRITE_SIZE(KiB)':>16s} {'bytes':>14s}")
for k, cs in sorted(tot.items()):
    if "giga" not in k: continue
    fs = cs["FETCH_SIZE"][0] / max(cs["FETCH_SIZE"][1], 1)
    wsz = cs["WRITE_SIZE"][0] / max(cs["WRITE_SIZE"][1], 1)
    print(f"{k[:90]:90s} {fs:16.1f} {fs * 1024 * 2:14.0f} {wsz:16.1f} {wsz * 1024:14.0f}")
PY
