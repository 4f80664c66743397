"""Per-kernel HBM traffic table from the two rocprofv3 PMC passes of tools/gpu_traffic.sh:
    python tools/make_traffic_table.py gpurun_out/traffic/<workload> out.json <workload>
bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024: FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide
coalesced read stream); that correction is calibrated for 16 B/lane streams, so the read side is an upper bound for kernels
with narrower loads.  bench.py looks kernels up by name fragment (roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import sys

src, dst, workload = sys.argv[1], sys.argv[2], sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else None          # steps of the workload the passes sampled (gpu_prof.py: iters + 1)
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        t = tot[r["Kernel_Name"]][r["Counter_Name"]]
        t[0] += float(r["Counter_Value"]); t[1] += 1
kernels = {}
print(f"== {workload}")
print(f"{'kernel':100s} {'FETCH_SIZE(KiB)':>16s} {'WRITE_SIZE(KiB)':>16s} {'bytes':>14s}")
for k, cs in sorted(tot.items()):
    if "giga" not in k:
        continue
    fs = cs["FETCH_SIZE"][0] / max(cs["FETCH_SIZE"][1], 1)
    wsz = cs["WRITE_SIZE"][0] / max(cs["WRITE_SIZE"][1], 1)
    kernels[k] = {"fetch_kib": round(fs, 1), "write_kib": round(wsz, 1), "bytes": int((2 * fs + wsz) * 1024),
                  "launches_sampled": cs["FETCH_SIZE"][1]}
    if steps:
        kernels[k]["launches_per_step"] = round(cs["FETCH_SIZE"][1] / steps, 2)
        kernels[k]["bytes_per_step"] = int(kernels[k]["bytes"] * cs["FETCH_SIZE"][1] / steps)
    print(f"{k[:100]:100s} {fs:16.1f} {wsz:16.1f} {kernels[k]['bytes']:14d}")
tab = {"_about": __doc__.strip(), "workload": workload + " (tools/gpu_prof.py), 32 scenes", "commit": os.environ.get("GIGA_COMMIT"),
       "kernels": kernels}
if steps:
    tab["steps_sampled"] = steps
    tab["bytes_per_step"] = sum(k["bytes_per_step"] for k in kernels.values())
    print(f"whole step: {tab['bytes_per_step'] / 1e6:.1f} MB of HBM traffic")
json.dump(tab, open(dst, "w"), indent=1)
