"""MFMA-pipe utilisation per kernel from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVES):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d out -o p -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu --single
    python3 tools/pmc_mfma.py out/p_results.db > profiles/rNN_mfma_util_<mode>.json

Reading (MI355X_MICROARCH.md, cycle constants + DVFS give-back): SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the
chip's 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so GRBM_GUI_ACTIVE / 8 = shader cycles of the dispatch and
GRBM_GUI_ACTIVE / 8 / duration = the clock the chip held.  utilisation = MFMA_BUSY / (cycles x 1024)."""
import collections
import json
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
t = lambda pre: [x for x in tabs if x.startswith(pre)][0]
q = (f"select s.kernel_name, i.name, p.value, d.start, d.end, d.id from {t('rocpd_pmc_event')} p "
     f"join {t('rocpd_info_pmc')} i on p.pmc_id = i.id "
     f"join {t('rocpd_kernel_dispatch')} d on p.event_id = d.event_id "
     f"join {t('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = set()
for name, ctr, val, st, en, did in c.execute(q):
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", re.sub(r"\(.*", "", name))[:60]
    agg[name][ctr] += float(val)
    if did not in seen:
        seen.add(did)
        agg[name]["ns"] += en - st
        agg[name]["launches"] += 1
out = {}
for name, v in sorted(agg.items(), key=lambda kv: -kv[1]["ns"])[:12]:
    cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0:
        continue
    out[name] = {"launches": int(v["launches"]), "ms": round(v["ns"] / 1e6, 3),
                 "clock_ghz": round(cyc / v["ns"], 3),
                 "mfma_pipe_utilisation": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0), 4),
                 }
print(json.dumps(out, indent=1))
