"""Retrieval evaluation on the device - mirror of reid/evaluate.py:33-105 (evaluate_all / evaluate / compute_mAP).

The reference loops over queries on the host: one GEMV, one D2H copy and one full argsort of the gallery per
query (evaluate.py:58-63).  Here one GEMM produces every similarity row and a rank-counting kernel finds, for
each good gallery item, its position among the non-junk items - no sort, no per-query copy.  Order of equal
scores: higher gallery index first (the reference's np.argsort default is not stable, so ties are undefined there).
"""
import numpy as np

from .engine import get_engine


def _np(a, dtype):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=dtype)


def evaluate_all(qf, ql, qc, gf, gl, gc, verbose=True, device=0):
    """Same signature and return convention as reid/evaluate.py:33: (CMC float32[ng], mAP float).
    Queries without any good match are skipped but still counted in the denominator (evaluate.py:49-50)."""
    qf, gf = _np(qf, np.float32), _np(gf, np.float32)
    ql, qc, gl, gc = (_np(a, np.int64) for a in (ql, qc, gl, gc))
    cmc_sum, ap, valid = get_engine(device).rank_eval(qf, ql, qc, gf, gl, gc)
    nq = qf.shape[0]
    total = 0.0
    for i in range(nq):              # python-float accumulation in query order, as the reference does
        if valid[i]:
            total += float(ap[i])
    cmc = cmc_sum.astype(np.float32) / nq
    mean_ap = total / nq
    if verbose:
        print('Rank@1:%f Rank@5:%f Rank@10:%f mAP:%f' % (cmc[0], cmc[4], cmc[9], mean_ap))
    try:
        import torch
        return torch.from_numpy(cmc), mean_ap
    except ImportError:              # numpy-only hosts
        return cmc, mean_ap
