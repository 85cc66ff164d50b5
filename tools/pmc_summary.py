"""Per-kernel PMC counter sums from a rocprofv3 --pmc results DB. usage: pmc_summary.py <db> [kernel substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
print("cols:", cols)
q = "select * from counters_collection"
rows = list(db.execute(q))
ki = cols.index("kernel_name") if "kernel_name" in cols else None
ci = cols.index("counter_name") if "counter_name" in cols else None
vi = cols.index("value") if "value" in cols else (cols.index("counter_value") if "counter_value" in cols else None)
di = cols.index("dispatch_id") if "dispatch_id" in cols else None
agg = {}
for r in rows:
    k = r[ki]
    if sub and sub not in k:
        continue
    import re
    m = re.search(r"(\w+_kernel)", k)
    k = m.group(1) if m else k.split("(")[0][-60:]
    e = agg.setdefault(k, {})
    e.setdefault(r[ci], []).append(r[vi])
for k, e in agg.items():
    print(k)
    for c, v in sorted(e.items()):
        print(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
