#!/usr/bin/env python
"""Per-kernel ISA statistics of a hipcc -save-temps gfx950 .s file: VGPR/AGPR use, scratch, MFMA and VALU mix.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -c X.hip -save-temps=obj -o /tmp/x.o ; python tools/isa_stats.py /tmp/X-hip-amdgcn-amd-amdhsa-gfx950.s [filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)\.Lfunc_end\d+:', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    def setv(k):
        r = re.search(r'\.set ' + re.escape(name) + r'\.' + k + r', (\d+)', s)
        return int(r.group(1)) if r else None
    pats = {"mfma": r"v_mfma", "fma_mix": r"v_fma_mix", "cvt_pk_f16": r"v_cvt_pk_f16_f32|v_cvt_pkrtz", "cvt_f16": r"v_cvt_f16_f32",
            "cvt_f32_f16": r"v_cvt_f32_f16", "max_i32": r"v_max_i32", "ds_read": r"ds_read", "ds_write": r"ds_write",
            "global_load": r"global_load", "scratch": r"scratch_", "s_nop": r"s_nop", "waitcnt": r"s_waitcnt", "accvgpr": r"v_accvgpr"}
    cnt = {k: len(re.findall(v, body)) for k, v in pats.items()}
    valu = len(re.findall(r'^\s+v_(?!mfma)', body, re.M))
    print(f"{name[:90]}\n   vgpr {setv('num_vgpr')} agpr {setv('num_agpr')} sgpr {setv('numbered_sgpr')} scratch {setv('private_seg_size')} "
          f"lines {body.count(chr(10))} valu {valu} " + " ".join(f"{k} {v}" for k, v in cnt.items() if v))
