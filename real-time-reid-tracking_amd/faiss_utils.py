"""Host mirror of the reference's retrieval helpers (reid/faiss_utils.py) on the HIP library: same names and argument
meaning, no faiss.  The k-NN search and the whole k-reciprocal re-ranking run on the device (csrc/rerank.hip).

* ``search_raw_array_pytorch(res, xb, xq, k)``  faiss_utils.py:56-118  (``res`` is accepted and ignored)
* ``search_index_pytorch(index, x, k)``         faiss_utils.py:30-53
* ``index_init_gpu / index_init_cpu``           faiss_utils.py:121-139  -> ``distance.IndexFlatL2``
* ``compute_jaccard_distance``                  faiss_utils.py:147-244
"""
import time

import numpy as np

from .distance import IndexFlatL2, _np, search_raw_array
from .engine import get_engine


def search_raw_array_pytorch(res, xb, xq, k, D=None, I=None, metric=None):
    """(D float32[nq,k] squared L2, I int32[nq,k]); ``res``/``metric`` kept for signature compatibility (L2 only)."""
    d, i = search_raw_array(xb, xq, k)
    if D is not None:
        D[...] = _like(D, d)
        d = D
    if I is not None:
        I[...] = _like(I, i)
        i = I
    return d, i


def search_index_pytorch(index, x, k, D=None, I=None):
    d, i = index.search(_np(x), k)
    if D is not None:
        D[...] = _like(D, d)
        d = D
    if I is not None:
        I[...] = _like(I, i)
        i = I
    return d, i


def _like(dst, src):
    try:
        import torch
        if isinstance(dst, torch.Tensor):
            return torch.from_numpy(np.ascontiguousarray(src)).to(dst.dtype)
    except ImportError:  # pragma: no cover
        pass
    return src


def index_init_gpu(ngpus, feat_dim):
    """One flat index on device 0 (the reference shards the base set over ``ngpus`` faiss indexes; multi-GPU gallery
    sharding here is ``parallel.knn_gallery_sharded``, one process per GPU)."""
    return IndexFlatL2(feat_dim)


def index_init_cpu(feat_dim):
    return IndexFlatL2(feat_dim)


def compute_jaccard_distance(target_features, k1=20, k2=6, print_flag=True, search_option=0, use_float16=False,
                             initial_rank=None, device=0):
    """k-reciprocal Jaccard distance of every pair of rows, float32 [N, N] numpy (as the reference returns).

    ``search_option`` selects a faiss back end in the reference; every option is the same brute-force squared-L2 search
    here.  ``use_float16`` (a numpy storage type in the reference) is accepted; the device computes in float32 and the
    result is cast at the end.  ``initial_rank`` (int [N, k1]) overrides the library's k-NN.
    """
    end = time.time()
    if print_flag:
        print('Computing jaccard distance...')
    out = get_engine(device).rerank_jaccard(_np(target_features), k1, k2, initial_rank)
    if use_float16:
        out = out.astype(np.float16)
    if print_flag:
        print("Jaccard distance computing time cost: {}".format(time.time() - end))
    return out
