#!/usr/bin/env python3
"""profiles/r05/c2_b4096_pmc.json from the three rocprofv3 --pmc passes of `bench.py --pmc-run` (tools/r05_measure.sh):
for EVERY decoder kernel of a C2 step (fused layer kernel, its k | v-only form, the embed + layer-0 q | k | v launch = SURVEY K1's
gather, the packed-sequence attention, the last layer's single-query attention) the median launch time, MFMA busy fraction, HBM
bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KB -> B: MI355X_MICROARCH.md's gfx950 correction) and the algorithmic bytes at
the packed row count of THAT run (printed by the profiled command itself), plus the decoder's HBM bytes per step (launches per
step x bytes per launch, summed).

usage: tools/r05_pmc.py <sq_dir> <fetch_dir> <write_dir> <rows.json> <out.json>"""
import collections
import csv
import glob
import json
import os
import re
import sys

D = 128
KERNELS = [  # label, regex on the kernel name, algorithmic HBM bytes per packed row of one launch (d = 128, float32), what they are
    ("layer", r"k_block_x6<0, 4, false", 4 * D * 6, "in: attention output + residual x; out: x' + the next layer's q | k | v"),
    ("layer_kv_only", r"k_block_x6<1, 4, false", 4 * D * 5, "the same, k | v only (feeds the rows-only last layer)"),
    ("embed_qkv0", r"k_block_x6<0, 4, true", 4 * D * 5, "K1: in: embedding row; out: x + layer 0's q | k | v"),
    ("attention", r"k_attn16h<", 4 * D * 4, "in: q | k | v rows; out: attention output (fragment-major)"),
    ("attention_last_row", r"k_attn_row32", 4 * D * 2, "in: k | v rows of every token; out: one row per sequence"),
    # the sequence-resident decoder (irs_set_decoder_seq; default from 1024 sequences up): ONE launch per step
    ("seq_decoder", r"k_block_x6<3, 8, false, 4, 2, true>", 4 * D * 2, "embedding + layers 0 .. n-2 with attention + the last layer's q|k|v and attention: "
     "in: embedding rows; out: x of the last fused layer (attention tiles and x' round trips between the layers are scratch traffic)"),
    ("seq_plan", r"k_plan_seq\(", 0, "workgroup plan of the sequence-resident launch (one workgroup; no row traffic)"),
]


def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[name][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, dur


def pick(agg, dur, rx):
    best = None
    for name in agg:
        if re.search(rx, name):
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
    nsteps = len(rows["packed_rows_per_step"])
    sq, sq_d = load(sq_dir)
    fe, fe_d = load(fetch_dir)
    wr, wr_d = load(write_dir)
    res = {"command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --pmc-run --steps 5 --warmup 2",
           "packed_rows_mean": R, "packed_rows_per_step": rows["packed_rows_per_step"], "users": rows["users"], "kernels": {}}
    step_bytes = step_alg = step_us = 0.0
    for label, rx, bpr, what in KERNELS:
        n_sq, n_fe, n_wr = pick(sq, sq_d, rx), pick(fe, fe_d, rx), pick(wr, wr_d, rx)
        if not (n_sq and n_fe and n_wr):
            continue
        m = {c: sum(v) / len(v) for c, v in sq[n_sq].items()}
        ds = sorted(sq_d[n_sq].values())
        med_us = ds[len(ds) // 2] / 1e3
        clk_cycles = m["SQ_BUSY_CYCLES"] / 32.0  # the counter sums over 32 shader engines
        fetch_kb = sum(fe[n_fe]["FETCH_SIZE"]) / len(fe[n_fe]["FETCH_SIZE"])
        write_kb = sum(wr[n_wr]["WRITE_SIZE"]) / len(wr[n_wr]["WRITE_SIZE"])
        k = {"kernel_name": n_sq, "what": what, "launches_profiled": len(ds), "launches_per_step": len(ds) / nsteps,
             "median_us": med_us, "mean_us": sum(ds) / len(ds) / 1e3, "clock_ghz": clk_cycles / med_us / 1e3,
             "mfma_busy": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * clk_cycles),  # 256 CUs x 4 SIMDs
             "FETCH_SIZE_kb_mean": fetch_kb, "WRITE_SIZE_kb_mean": write_kb,
             "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
             "algorithmic_bytes_per_launch": bpr * R, "algorithmic_bytes_per_packed_row": bpr}
        k["algorithmic_gbs"] = k["algorithmic_bytes_per_launch"] / (med_us * 1e-6) / 1e9
        k["frac_hbm_peak"] = k["algorithmic_gbs"] / 8000.0
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c, lab in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_ANY", "active")):
                if c in m:
                    k[lab] = m[c] / wc
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU"):
            if c in m:
                k[c] = m[c]
        k["traffic_over_algorithmic"] = k["hbm_bytes_per_launch"] / k["algorithmic_bytes_per_launch"] if bpr else None
        res["kernels"][label] = k
        step_bytes += k["launches_per_step"] * k["hbm_bytes_per_launch"]
        step_alg += k["launches_per_step"] * k["algorithmic_bytes_per_launch"]
        step_us += k["launches_per_step"] * med_us
    res["decoder_hbm_bytes_per_step"] = step_bytes
    res["decoder_algorithmic_bytes_per_step"] = step_alg
    res["decoder_kernel_us_per_step"] = step_us
    if "layer" in res["kernels"]:
        res["kernels"]["k_block_x6"] = res["kernels"]["layer"]  # (the key earlier rounds' readers use)
    json.dump(res, open(out, "w"), indent=1)
    for p, k in res["kernels"].items():
        if p == "k_block_x6":
            continue
        print(f"{p:20s} x{k['launches_per_step']:.0f}/step  med {k['median_us']:8.1f} us  mfma_busy {100 * k['mfma_busy']:5.1f} %  clk {k['clock_ghz']:.2f} GHz  "
              f"HBM {k['hbm_bytes_per_launch'] / 1e9:.3f} GB vs algorithmic {k['algorithmic_bytes_per_launch'] / 1e9:.3f} GB "
              f"(x{(k['traffic_over_algorithmic'] or 0.0):.2f}) = {k['algorithmic_gbs']:.0f} GB/s of algorithmic bytes ({k['frac_hbm_peak']:.2f} of 8 TB/s)")
    print(f"decoder HBM bytes per step {step_bytes / 1e9:.2f} GB (algorithmic {step_alg / 1e9:.2f} GB), kernel time {step_us / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
