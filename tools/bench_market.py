"""BASELINE config 5 at full size on one MI355X: query(3368) x gallery(15913) x 512 distance matrix, fused row arg-min
(rank-1 index), CMC/mAP by rank counting, 20-NN of the gallery - against the CPU oracle.  python tools/bench_market.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import matching
from reid_amd import _ffi, synth
from reid_amd.engine import get_engine
from reid_amd.evaluate import evaluate_all

eng = get_engine(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
eng.set_stream(stream.cuda_stream)
qf, ql, qc, gf, gl, gc = synth.clustered_embeddings(3368, 15913, d=512, n_ids=751, n_cams=6, seed=4, sigma=1.6)
dq, dg = torch.from_numpy(qf).cuda(), torch.from_numpy(gf).cuda()
dist = torch.empty((3368, 15913), dtype=torch.float32, device="cuda")
idx = torch.empty(3368, dtype=torch.int32, device="cuda")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


dist_ms = timed(lambda: eng.distmat_dev(dq.data_ptr(), 3368, dg.data_ptr(), 15913, 512, _ffi.METRIC_L2, dist.data_ptr()))
top1_ms = timed(lambda: eng.argmin_rows_dev(dq.data_ptr(), 3368, dg.data_ptr(), 15913, 512, _ffi.METRIC_L2, idx.data_ptr()))
torch.cuda.synchronize()
t0 = time.perf_counter()
cmc, mAP = evaluate_all(qf, ql, qc, gf, gl, gc, verbose=False)
eval_ms = (time.perf_counter() - t0) * 1e3          # includes H2D of both sets
t0 = time.perf_counter()
D, I = eng.knn(gf[:4096], gf, 20)
knn_ms = (time.perf_counter() - t0) * 1e3
# CPU oracle side by side (numpy BLAS on the host cores)
t0 = time.perf_counter()
dref = matching.euclidean_dist(qf, gf)
cpu_dist_ms = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
cmc_r, map_r = matching.evaluate_all(qf[:256], ql[:256], qc[:256], gf, gl, gc)
cpu_eval_ms_256 = (time.perf_counter() - t0) * 1e3
cmc256, map256 = evaluate_all(qf[:256], ql[:256], qc[:256], gf, gl, gc, verbose=False)
top1 = idx.cpu().numpy()
srt = np.partition(dref, 1, axis=1)[:, :2]
decided = (srt[:, 1] - srt[:, 0]) > 1e-5
out = {"workload": "BASELINE configs[4] on 1 GPU: 3368 x 15913 x 512", "distmat_ms": round(dist_ms, 3),
       "distmat_tflops": round(2 * 3368 * 15913 * 512 / dist_ms / 1e9, 1), "distmat_gbs": round(4 * (3368 * 512 + 15913 * 512 + 3368 * 15913) / dist_ms / 1e6, 1),
       "rank1_argmin_ms": round(top1_ms, 3), "evaluate_all_ms_incl_h2d": round(eval_ms, 1), "rank1": float(cmc[0]), "mAP": float(mAP),
       "knn20_4096x15913_ms_incl_copies": round(knn_ms, 1),
       "top1_equal_where_decided": bool((top1[decided] == dref.argmin(1)[decided]).all()), "decided": int(decided.sum()),
       "cmc_equal_oracle_256q": bool((np.asarray(cmc256) == cmc_r).all()), "map_abs_err_256q": abs(map256 - map_r),
       "cpu_distmat_ms": round(cpu_dist_ms, 1), "cpu_evaluate_all_ms_per_256_queries": round(cpu_eval_ms_256, 1)}
print(json.dumps(out))
