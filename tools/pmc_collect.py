#!/usr/bin/env python3
"""Development: mean counter values per attention kernel from rocprofv3 --pmc output directories.  python tools/pmc_collect.py <dir> [...]"""
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
for d in sys.argv[1:]:
    acc = {}
    for cc in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(cc)):
            if "attn_fwd_kernel" not in row["Kernel_Name"]: continue
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    if not acc: print(d, "no data"); continue
    m = {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in acc.items()}   # second half of the launches: clocks settled
    g = m.get("GRBM_GUI_ACTIVE", float("nan")) / 8
    line = f"{os.path.basename(d.rstrip('/')):14s} cycles/launch {g / 1e3:8.1f}k"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m: line += f"  mfma busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * g):.3f}"
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        line += f"  wave-cycles {w / 1e6:.1f}M  wait_any {m.get('SQ_WAIT_ANY', 0) / w:.3f}  wait_inst {m.get('SQ_WAIT_INST_ANY', 0) / w:.3f}  active_inst {m.get('SQ_ACTIVE_INST_ANY', 0) / w:.3f}"
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
        if k in m: line += f"  {k[3:]} {m[k] / 1e6:.2f}M"
    print(line)
