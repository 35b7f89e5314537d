"""Per-kernel totals from a rocprofv3 --kernel-trace database (rocpd sqlite): python tools/rocprof_summary.py <db> [top]"""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id").fetchall()
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)
    n = re.sub(r'\(.*', '', n)
    agg[n][0] += 1
    agg[n][1] += e - s
tot = sum(v[1] for v in agg.values())
print("kernel,calls,total_ms,avg_us,percent")
for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
    print(f"{n[:90]},{k},{t / 1e6:.3f},{t / k / 1e3:.1f},{100 * t / tot:.1f}")
