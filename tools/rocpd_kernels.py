"""Per-kernel durations out of a rocprofv3 rocpd database (the default output format of ROCm 7.2).
usage: python tools/rocpd_kernels.py <results.db> [marker-kernel-substring]
With a marker (a kernel launched once per step, e.g. k_path_step) the total is also given per step."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 from {kd} d "
     f"join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc")
rows = list(c.execute(q))
tot = sum(r[3] for r in rows)
for r in rows[:24]:
    print(f"{r[0][:64]:64s} n={r[1]:7d} avg={r[2]:9.2f} us  {100 * r[3] / tot:5.1f}%")
if len(sys.argv) > 2:
    hit = [r[1] for r in rows if sys.argv[2] in r[0]]
    if hit:
        print(f"kernel time per {sys.argv[2]}: {tot / hit[0]:.1f} us")
    else:
        print(f"(no kernel named *{sys.argv[2]}* in this trace: no per-step figure)")
