"""Large k-NN: candidates on the f16 matrix pipe + exact fp32 refinement (knn_wide.hip) against the fused fp32 search
(dist_select.hip): results must be bit-identical; times of both.  python tools/knn_wide_check.py [N] [D] [k]"""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi, parallel, synth
from reid_amd._ffi import check
from reid_amd.engine import get_engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 19281
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1263
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
eng = get_engine(0)
from reid_amd import weights
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])     # the library's zero page lives with the weights
sw = _ffi.debug_lib().reid_debug_knn_wide
sw.restype = C.c_int
sw.argtypes = [C.c_void_p, C.c_int, C.c_int]
_, _, _, x, _, _ = synth.clustered_embeddings(1, N, d=D, n_ids=751, n_cams=6, seed=5, sigma=0.9)
x[N // 2] = x[7]                       # duplicate rows: exact ties, lowest index first
dx = parallel.DevArray.from_numpy(eng, x)
dD = parallel.DevArray(eng, (N, K))
dI = parallel.DevArray(eng, (N, K), np.int32)


def run(enable, force, reps=3):
    check(sw(eng.h, enable, force))
    eng.knn_dev(dx.ptr, N, dx.ptr, N, D, K, dD.ptr, dI.ptr)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.knn_dev(dx.ptr, N, dx.ptr, N, D, K, dD.ptr, dI.ptr)
    eng.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    return dD.numpy().copy(), dI.numpy().copy(), ms


D0, I0, t0 = run(0, 0)
D1, I1, t1 = run(1, 0)
D2, I2, t2 = run(1, 97, reps=1)
print("N %d D %d k %d: fused fp32 %.2f ms (%.0f TF/s), wide %.2f ms (%.0f TF/s algorithmic), wide with every 97th row forced through the exact fallback %.2f ms"
      % (N, D, K, t0, 2.0 * N * N * D / t0 / 1e9, t1, 2.0 * N * N * D / t1 / 1e9, t2))
print("indices equal:", bool(np.array_equal(I0, I1)), " distances bit-equal:", bool(np.array_equal(D0, D1)),
      " | forced fallback: indices", bool(np.array_equal(I0, I2)), "distances", bool(np.array_equal(D0, D2)))
if not np.array_equal(I0, I1):
    bad = np.flatnonzero((I0 != I1).any(1))
    print("rows that differ:", len(bad), bad[:10])
    r = bad[0]
    print(I0[r], I1[r], D0[r], D1[r])
elif not np.array_equal(D0, D1):
    bad = np.argwhere(D0 != D1)
    print("distance mismatches:", len(bad), bad[:5], D0[tuple(bad[0])], D1[tuple(bad[0])])
check(sw(eng.h, 1, 0))
