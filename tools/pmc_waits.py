"""Per-kernel wait / issue breakdown from one rocprofv3 PMC pass:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d out -o p -- python3 tools/time_pass.py 2 1024
    python3 tools/pmc_waits.py out/.../p_results.db
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles, summed over waves); MFMA_BUSY in cycles summed over SIMDs."""
import collections, re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
t = lambda pre: [x for x in tabs if x.startswith(pre)][0]
q = (f"select s.kernel_name, i.name, p.value, d.start, d.end, d.id from {t('rocpd_pmc_event')} p "
     f"join {t('rocpd_info_pmc')} i on p.pmc_id = i.id join {t('rocpd_kernel_dispatch')} d on p.event_id = d.event_id "
     f"join {t('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = set()
for name, ctr, val, st, en, did in c.execute(q):
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", re.sub(r"\(.*", "", name))[:64]
    agg[name][ctr] += float(val)
    if did not in seen:
        seen.add(did); agg[name]["ns"] += en - st; agg[name]["launches"] += 1
for name, v in sorted(agg.items(), key=lambda kv: -kv[1]["ns"])[:10]:
    cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0: continue
    wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    print("%-64s %3d launches %8.3f ms  clock %.2f GHz  mfma util %.3f  | of wave cycles: wait_any %.2f  wait_inst %.2f (lds %.2f)  active %.2f | lds bank conflict cycles/busy %.3f"
          % (name, v["launches"], v["ns"] / 1e6, cyc / v["ns"], v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024), v.get("SQ_WAIT_ANY", 0) / wc,
             v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_WAIT_INST_LDS", 0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_BUSY_CYCLES", 1), 1)))
