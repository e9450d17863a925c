"""Per-kernel totals from a rocprofv3 rocpd (.db) kernel trace:  python tools/rocpd_kernel_stats.py results.db [out.csv] [steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
rows = db.execute(f"select s.{name_col}, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
agg = {}
for n, dt in rows:
    n = re.sub(r"\(.*", "", n).replace("(anonymous namespace)::", "").replace("void ", "")
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += dt
tot = sum(a[1] for a in agg.values())
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
lines = ["kernel,calls,total_ms,avg_us,percent,ms_per_step"]
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f'"{n}",{c},{t / 1e6:.3f},{t / c / 1e3:.2f},{100 * t / tot:.2f},{t / 1e6 / steps:.3f}')
out = "\n".join(lines)
if len(sys.argv) > 2 and sys.argv[2] != "-":
    open(sys.argv[2], "w").write(out + "\n")
print("\n".join(lines[:40]))
print(f"TOTAL {tot / 1e6:.1f} ms over {steps} steps = {tot / 1e6 / steps:.1f} ms/step")
