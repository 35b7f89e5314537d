"""N x M distances on the device - mirrors of reid/losses/utils.py:12-35 and the brute-force k-NN of
reid/faiss_utils.py:56-139.  Inputs may be numpy arrays or torch tensors (CPU or GPU); outputs are numpy."""
import numpy as np

from . import _ffi
from .engine import get_engine


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def euclidean_dist(x, y, device=0):
    """sqrt(clamp(|x|^2 + |y|^2 - 2 x.y^T, 1e-12)) - reid/losses/utils.py:21-35."""
    return get_engine(device).distmat(_np(x), _np(y), _ffi.METRIC_L2)


def cosine_dist(x, y, device=0):
    """(1 - cos)/2 - reid/losses/utils.py:12-18."""
    return get_engine(device).distmat(_np(x), _np(y), _ffi.METRIC_COS_HALF)


def cosine_distance(x, y, device=0):
    """1 - cos on L2-normalised rows: DeepSORT's appearance cost, gated by MAX_DIST 0.15 (deep_sort.yaml:3)."""
    return get_engine(device).distmat(_np(x), _np(y), _ffi.METRIC_COS)


def nearest(x, y, metric=_ffi.METRIC_L2, device=0):
    """(argmin index int32[m], min value float32[m]) per row of the distance matrix; ties -> lowest index."""
    return get_engine(device).argmin_rows(_np(x), _np(y), metric)


def search_raw_array(xb, xq, k, device=0):
    """Brute-force squared-L2 k-NN, argument order of search_raw_array_pytorch(res, xb, xq, k)
    (reid/faiss_utils.py:56): returns (D float32[nq,k], I int32[nq,k])."""
    return get_engine(device).knn(_np(xq), _np(xb), k)


class IndexFlatL2:
    """The slice of faiss.IndexFlatL2 the reference uses (reid/faiss_utils.py:138-139,176-181): add + search."""

    def __init__(self, d, device=0):
        self.d, self.device = int(d), device
        self._xb = np.empty((0, self.d), np.float32)

    @property
    def ntotal(self):
        return self._xb.shape[0]

    def reset(self):
        self._xb = np.empty((0, self.d), np.float32)

    def add(self, x):
        x = _np(x)
        assert x.ndim == 2 and x.shape[1] == self.d
        self._xb = np.concatenate([self._xb, x], 0)

    def search(self, x, k):
        D, I = get_engine(self.device).knn(_np(x), self._xb, k)
        return D, I.astype(np.int64)
