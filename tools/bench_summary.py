#!/usr/bin/env python3
"""Print the numbers of a bench.py JSON line that a build loop reads first.  usage: tools/bench_summary.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("C2 value %.4g %s  ms/step %.3f  verified %s" % (d["value"], d["unit"], d["ms_per_step"], d.get("verified")))
r = d.get("roofline") or {}
print("roofline", {k: r.get(k) for k in ("bound", "achieved", "peak", "frac", "frac_hbm", "frac_mfma", "traffic")})
print("families", r.get("family_ms_per_step"))
for k in ("step_roofline",):
    if k in d: print(k, d[k])
print("lat b1 %s ms, b128 %s, b1024 %s" % (d.get("path_gen_p50_ms_b1"), d.get("path_gen_ms_per_user_b128"), d.get("path_gen_ms_per_user_b1024")))
for leg in ("c3_1M_items", "c4_item_sharded"):
    x = d.get(leg)
    if x: print(leg, "ms/step %.3f value %.4g verified %s rowdiff %s phases %s" % (x["ms_per_step"], x["value"], x.get("verified"), x.get("max_row_diff"), x.get("phase_ms_rank0")))
c5 = (d.get("c4_item_sharded") or {}).get("c5_beam32")
if c5: print("c5", {k: v for k, v in c5.items() if "ms" in k})
sc = d.get("scoring")
if sc: print("scoring", json.dumps(sc)[:1500])
print("cpu_baseline", d.get("cpu_baseline"))
if "extras_error" in d: print("EXTRAS ERROR", d["extras_error"])
