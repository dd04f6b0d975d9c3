#!/usr/bin/env python3
"""profiles/r04/c2_b4096_pmc.json from the three rocprofv3 --pmc passes of `bench.py --pmc-run` (tools/r04_measure.sh):
per decoder kernel (k_block, k_attn16, k_embed_qkv) the median launch time, MFMA busy fraction, HBM bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE, KB -> B: MI355X_MICROARCH.md's gfx950 correction) and the algorithmic bytes at the packed
row count of THAT run (printed by the profiled command itself).

usage: tools/r04_pmc.py <sq_dir> <fetch_dir> <write_dir> <rows.json> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = {  # name prefix -> algorithmic HBM bytes per packed row of one launch (d = 128, float32)
    "k_block_x6": 4 * 128 * (2 + 1 + 3),   # in: attention output + residual x; out: x' and the next layer's q | k | v
    "k_block": 4 * 128 * (2 + 1 + 3),      # (IRS_GEMM_F32: the same kernel on float32 MFMAs; with IRS_GEMM_X6 only the last layer's, whose outputs differ)
    "k_attn16": 4 * 128 * (3 + 1),         # in: q | k | v; out: attention output (fragment-major)
    "k_embed_qkv": 4 * 128 * (1 + 1 + 3),  # in: embedding row; out: x and layer 0's q | k | v
}


def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[name][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, dur


def pick(agg, dur, prefix):
    """the instantiation of `prefix` with the largest total time in this pass"""
    best = None
    for name in agg:
        base = name.replace("void ", "")
        if base.startswith(prefix + "<") or base.startswith(prefix + "(") or base == prefix:
            tot = sum(dur[name].values())
            if best is None or tot > best[1]:
                best = (name, tot)
    return best[0] if best else None


def main():
    sq_dir, fetch_dir, write_dir, rows_json, out = sys.argv[1:6]
    rows = None
    for line in open(rows_json):
        line = line.strip()
        if line.startswith("{") and "pmc_run" in line:
            rows = json.loads(line)
    assert rows, "no pmc_run line in " + rows_json
    R = rows["packed_rows_mean"]
    sq, sq_d = load(sq_dir)
    fe, fe_d = load(fetch_dir)
    wr, wr_d = load(write_dir)
    res = {"command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --pmc-run --steps 5 --warmup 2",
           "packed_rows_mean": R, "packed_rows_per_step": rows["packed_rows_per_step"], "users": rows["users"], "kernels": {}}
    for prefix, bpr in KERNELS.items():
        n_sq, n_fe, n_wr = pick(sq, sq_d, prefix), pick(fe, fe_d, prefix), pick(wr, wr_d, prefix)
        if not (n_sq and n_fe and n_wr):
            continue
        m = {c: sum(v) / len(v) for c, v in sq[n_sq].items()}
        ds = sorted(sq_d[n_sq].values())
        med_us = ds[len(ds) // 2] / 1e3
        clk_cycles = m["SQ_BUSY_CYCLES"] / 32.0  # the counter sums over 32 shader engines
        fetch_kb = sum(fe[n_fe]["FETCH_SIZE"]) / len(fe[n_fe]["FETCH_SIZE"])
        write_kb = sum(wr[n_wr]["WRITE_SIZE"]) / len(wr[n_wr]["WRITE_SIZE"])
        k = {"kernel_name": n_sq, "launches_profiled": len(ds), "median_us": med_us, "mean_us": sum(ds) / len(ds) / 1e3,
             "clock_ghz": clk_cycles / med_us / 1e3,
             "mfma_busy": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * clk_cycles),  # 256 CUs x 4 SIMDs
             "FETCH_SIZE_kb_mean": fetch_kb, "WRITE_SIZE_kb_mean": write_kb,
             "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
             "algorithmic_bytes_per_launch": bpr * R, "algorithmic_bytes_per_packed_row": bpr}
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c, lab in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_ANY", "active")):
                if c in m:
                    k[lab] = m[c] / wc
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F32"):
            if c in m:
                k[c] = m[c]
        k["traffic_over_algorithmic"] = k["hbm_bytes_per_launch"] / k["algorithmic_bytes_per_launch"]
        res["kernels"][prefix] = k
    json.dump(res, open(out, "w"), indent=1)
    for p, k in res["kernels"].items():
        print(f"{p:14s} med {k['median_us']:8.1f} us  mfma_busy {100 * k['mfma_busy']:5.1f} %  clk {k['clock_ghz']:.2f} GHz  "
              f"HBM {k['hbm_bytes_per_launch'] / 1e9:.3f} GB vs algorithmic {k['algorithmic_bytes_per_launch'] / 1e9:.3f} GB "
              f"(x{k['traffic_over_algorithmic']:.2f})")


if __name__ == "__main__":
    main()
