"""Kernel timeline of one forward out of a rocprofv3 --kernel-trace database:
    python tools/timeline.py <db> <launches per forward | first kernel of a forward> [which, counted from the end]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
per = sys.argv[2]      # launches per forward, or the name of the kernel every forward starts with
which = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = c.execute(f"select s.kernel_name,d.start,d.end,d.grid_size_x,d.workgroup_size_x,d.grid_size_y from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
if per.isdigit():
    per = int(per)
    rows = rows[-per * which:len(rows) - per * (which - 1)]
else:
    starts = [i for i, r in enumerate(rows) if per in r[0]]
    rows = rows[starts[-which]:starts[-which + 1] if which > 1 else len(rows)]
prev, t0 = None, rows[0][1]
for n, s, e, g, w, gy in rows:
    n = re.sub(r'\(.*', '', n)
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)[:58]
    print("%-60s %7.1f us gap %5.1f  grid %5d x %d  block %d  @%.1f" % (n, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, g // w, gy, w, (s - t0) / 1e3))
    prev = e
print("span %.1f us" % ((rows[-1][2] - rows[0][1]) / 1e3))
