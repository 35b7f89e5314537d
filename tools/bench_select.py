"""Fused distance + selection (dist_select.hip) against the two-pass form (distance matrix + topk_rows / argmin_rows) at the
BASELINE sizes: Market-size search 3368 x 15913 x 512 (arg-min and top-20) and 4096 x 4096 x 512 (arg-min).
python tools/bench_select.py            (REID_DEBUG_SWITCHES=select_two_pass=1 python tools/bench_select.py  for the two-pass numbers)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi, parallel, synth
from reid_amd.engine import get_engine

eng = get_engine(0)
eng.debug_switches_from_env()
out = {"two_pass": eng.debug_switch("select_two_pass")}
for tag, m, n, k in (("market_top20", 3368, 15913, 20), ("market_argmin", 3368, 15913, 1), ("sq4096_argmin", 4096, 4096, 1),
                     ("gallery_self_top20", 15913, 15913, 20)):
    qf, _, _, gf, _, _ = synth.clustered_embeddings(m, n, d=512, n_ids=751, n_cams=6, seed=4, sigma=3.0)
    dq, dg = parallel.DevArray.from_numpy(eng, qf), parallel.DevArray.from_numpy(eng, gf)
    dD, dI = parallel.DevArray(eng, (m, k)), parallel.DevArray(eng, (m, k), np.int32)
    fn = (lambda: eng.knn_dev(dq.ptr, m, dg.ptr, n, 512, k, dD.ptr, dI.ptr)) if k > 1 else \
         (lambda: eng.argmin_rows_dev(dq.ptr, m, dg.ptr, n, 512, _ffi.METRIC_L2, dI.ptr, dD.ptr))
    for _ in range(3):
        fn()
    eng.sync()
    eng.timer_start()
    for _ in range(10):
        fn()
    ms = eng.timer_stop() / 10
    out[tag] = {"ms": round(ms, 4), "tflops": round(2.0 * m * n * 512 / ms / 1e9, 1)}
    for a in (dq, dg, dD, dI):
        a.free()
print(json.dumps(out))
