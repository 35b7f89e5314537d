"""Device busy / idle time from a rocprofv3 --kernel-trace database: python tools/gpu_idle.py <db> [skip_first_ms]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
rows = c.execute(f"select start, end from {kd} order by start").fetchall()
n = len(rows)
rows = rows[n // 4:]                      # steady state: skip set-up and warm-up
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ksum = sum(e - s for s, e in rows)
big = [g for g in gaps if g > 20000]
print("kernels %d  span %.1f ms  busy (union) %.1f ms = %.1f %%  sum of durations %.1f ms  idle gaps > 20 us: %d totalling %.1f ms"
      % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, ksum / 1e6, len(big), sum(big) / 1e6))
