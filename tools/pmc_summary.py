"""Print per-kernel PMC counter averages from a rocprofv3 rocpd database (last dispatches of each kernel)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
cur = c.execute("select * from pmc_events limit 1")
cols = [d[0] for d in cur.description]
# columns vary across rocprofiler versions: find the useful ones
q = "select * from pmc_events"
rows = list(c.execute(q))
ix = {n: i for i, n in enumerate(cols)}
name_i = ix.get("name", ix.get("kernel_name"))
cnt_i = ix.get("counter_name", ix.get("pmc_name", ix.get("symbol")))
val_i = ix.get("value", ix.get("counter_value"))
if name_i is None or cnt_i is None or val_i is None:
    print("columns:", cols)
    print(rows[:3])
    sys.exit(0)
agg = {}
for r in rows:
    if flt not in str(r[name_i]):
        continue
    key = (str(r[name_i])[:60], r[cnt_i])
    a = agg.setdefault(key, [0.0, 0])
    a[0] += float(r[val_i])
    a[1] += 1
for (k, cn), (s, n) in sorted(agg.items()):
    print(f"{k:60s} {cn:32s} avg={s / n:16.1f} n={n}")
