#!/usr/bin/env python3
"""Derived figures from profiles/<name>/pmc_summary.json + kernel_stats.csv (written by tools/summarize_profile.py), one line per
attention kernel instantiation, as DESIGN.md section 4.3 quotes them:
  MFMA busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)   (share of all SIMD cycles the matrix pipe is busy)
  VALU, SALU, LDS per MFMA = SQ_INSTS_x / SQ_INSTS_MFMA   (whole launch: prologue, Q quantisation, rescues and epilogue included)
  LDS active  = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CYCLES-equivalent is not comparable across passes, so only the conflict share is printed
usage: python tools/pmc_table.py profiles/r03_c2 [profiles/r03_c3 ...]   (markdown on stdout)
"""
import csv
import json
import os
import sys

print("| profile | kernel (template arguments) | avg µs (rocprofv3 --stats) | MFMA busy | VALU / MFMA | SALU / MFMA | LDS / MFMA | wait-inst share | LDS bank conflicts | HBM bytes / launch (PMC) | algorithmic |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for d in sys.argv[1:]:
    pm = json.load(open(os.path.join(d, "pmc_summary.json")))
    avg = {}
    with open(os.path.join(d, "kernel_stats.csv")) as f:
        for row in csv.DictReader(f):
            avg[row["Name"].split("(")[0][:90]] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
    tr = json.load(open(os.path.join(d, "traffic.json"))) if os.path.exists(os.path.join(d, "traffic.json")) else {}
    for k, v in pm.items():
        if "attn" not in k or "fwd" not in k or not v.get("SQ_INSTS_MFMA"):
            continue
        g = lambda n: v.get(n, float("nan"))
        busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g("GRBM_GUI_ACTIVE") / 8)
        us = avg.get(k, (float("nan"), 0))
        mine = tr.get("kernel") == k
        print(f"| {os.path.basename(d.rstrip('/'))} | `{k[k.index('qattn::') + 7:]}` | {us[0]:.1f} ({us[1]} calls) | {busy:.3f} | {g('SQ_INSTS_VALU') / g('SQ_INSTS_MFMA'):.2f} | "
              f"{g('SQ_INSTS_SALU') / g('SQ_INSTS_MFMA'):.2f} | {g('SQ_INSTS_LDS') / g('SQ_INSTS_MFMA'):.2f} | {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f} | "
              f"{g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.4f} | {(g('FETCH_SIZE') * 2048 + g('WRITE_SIZE') * 1024) / 1e6:.1f} MB | "
              f"{tr.get('algorithmic_bytes', 0) / 1e6 if mine else float('nan'):.1f} MB |")
