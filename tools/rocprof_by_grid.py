"""Median duration per (kernel, grid size) from a rocprofv3 --kernel-trace database: python tools/rocprof_by_grid.py <db>"""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
agg = collections.defaultdict(list)
for n, s, e, g, w in rows:
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)[:44]
    agg[(n, g // w)].append((e - s) / 1e3)
print("kernel,blocks,calls,median_us")
for k, v in sorted(agg.items()):
    v = sorted(v)
    print("%s,%d,%d,%.1f" % (k[0], k[1], len(v), v[len(v) // 2]))
