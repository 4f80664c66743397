"""profiles/r01_traffic_c2.json from the table printed by tools/gpu_traffic.sh (gpurun_out/final/traffic.txt)."""
import json
import re
import sys

STAGE_OF = {   # kernel-name fragment -> stage names of bench.py that run this instantiation
    "convin_project_kernel<float,": ["convin_project"],
    "plane_finalize_kernel<float>": ["plane_finalize"],
    "conv16_kernel<float, 0, 32, 0, 32, 40, 40, 2, false, true>": ["unet.down0.conv1", "unet.up1.conv2"],
    "conv16_kernel<float, 0, 32, 0, 32, 40, 40, 2, true, true>": ["unet.down0.conv2+pool"],
    "conv16_kernel<float, 0, 32, 0, 64, 20, 20, 1, false, true>": ["unet.down1.conv1"],
    "conv16_kernel<float, 0, 64, 0, 64, 20, 20, 1, true, true>": ["unet.down1.conv2+pool"],
    "conv16_kernel<float, 0, 64, 0, 128, 10, 10, 1, false, true>": ["unet.down2.conv1"],
    "conv16_kernel<float, 0, 128, 0, 128, 10, 10, 1, false, true>": ["unet.down2.conv2"],
    "conv16_kernel<float, 1, 128, 0, 64, 10, 10, 2, false, false>": ["unet.up0.upconv"],
    "conv16_kernel<float, 0, 64, 64, 64, 20, 20, 1, false, true>": ["unet.up0.conv1"],
    "conv16_kernel<float, 0, 64, 0, 64, 20, 20, 1, false, true>": ["unet.up0.conv2"],
    "conv16_kernel<float, 1, 64, 0, 32, 20, 20, 2, false, false>": ["unet.up1.upconv"],
    "conv16_kernel<float, 0, 32, 32, 32, 40, 40, 2, false, true>": ["unet.up1.conv1"],
    "conv16_kernel<float, 2, 32, 0, 32, 40, 40, 2, false, false>": ["unet.conv_final"],
    "decoder_f32_kernel<2, false>": ["decoder_occ"],
}

src, dst = sys.argv[1], sys.argv[2]
stages = {}
for line in open(src):
    m = re.match(r"^(void giga::.*?)\s+([\d.]+)\s+(\d+)\s+([\d.]+)\s+(\d+)\s*$", line)
    if not m:
        continue
    name, fk, _, wk, _ = m.groups()
    for frag, names in STAGE_OF.items():
        if frag in name:
            for st in names:
                stages[st] = {"kernel": name.strip(), "fetch_kib": float(fk), "write_kib": float(wk),
                              "bytes": int((2 * float(fk) + float(wk)) * 1024)}
out = {"_about": "HBM traffic per launch of the c2 step (B=32, fp32) from rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in separate "
                 "--pmc passes (tools/gpu_traffic.sh). Units KiB. bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE is "
                 "doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read stream); that correction is "
                 "calibrated for 16 B/lane streams only, so the read side is an upper bound for kernels with narrower loads.",
       "stages": stages}
json.dump(out, open(dst, "w"), indent=1)
print(len(stages), "stages")
