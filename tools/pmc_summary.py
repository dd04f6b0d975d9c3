#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per (kernel, grid): median duration and
SQ ratios.  usage: tools/pmc_summary.py <dir-or-csv> [substring ...]"""
import collections, csv, glob, os, sys
path = sys.argv[1]
subs = sys.argv[2:] or [""]
f = path if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"][:40], r["Grid_Size"])
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    if not any(s in k[0] for s in subs):
        continue
    d = sorted(dur[k])[len(dur[k]) // 2] / 1e3
    m = {c: sum(x) / len(x) for c, x in v.items()}
    out = [f"{k[0]:40s} grid={k[1]:>9s} n={len(dur[k])//max(len(m),1):3d} med_us={d:8.1f}"]
    wc = m.get("SQ_WAVE_CYCLES")
    if "SQ_BUSY_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        clk_cycles = m["SQ_BUSY_CYCLES"] / 32.0  # summed over 32 shader engines
        out.append(f"clk~{clk_cycles / d / 1e3:.2f}GHz mfma_busy={100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * clk_cycles):.0f}%")
    if wc:
        for c, lab in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_ANY", "active")):
            if c in m:
                out.append(f"{lab}={100 * m[c] / wc:.0f}%")
    if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"] > 0:
        out.append(f"lds_conf={m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.2f}")
    for c in m:
        if c not in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                     "SQ_ACTIVE_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
            out.append(f"{c}={m[c]:.3g}")
    print(" ".join(out))
