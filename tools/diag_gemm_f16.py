"""Where do the cycles of one K-loop iteration go?  Runs the s_memtime-stamped diagnostic build of the fp16 GEMM loop."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import get_engine
from reid_amd import _ffi
from reid_amd._ffi import check

eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
gf = _ffi.debug_lib().reid_debug_gemm_f16
gf.restype = C.c_int
gf.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.POINTER(C.c_float), C.c_void_p]
for cfg, bk in ((3256642, 64), (3128643, 64), (3128642, 64), (3256324, 32)):
    m, n, k = 8192, 8192, 4096
    diag = np.zeros((64, 8, 4), np.uint64)
    ms = C.c_float()
    check(gf(eng.h, m, n, k, cfg, 3, C.byref(ms), diag.ctypes.data_as(C.c_void_p)))
    nk = k // bk
    d = diag.astype(np.float64) / nk
    tot = d.sum(2)
    print("cfg %d: %.0f TF (stamped build); per K-tile cycles (mean over waves): wait %.0f  barrier %.0f  dma-issue %.0f  reads+mfma %.0f  total %.0f"
          % (cfg, 2.0 * m * n * k / (ms.value * 1e-3) / 1e12, d[..., 0].mean(), d[..., 1].mean(), d[..., 2].mean(), d[..., 3].mean(), tot.mean()))
    print("     waves 0-3 vs 4-7 total:", tot[:, :4].mean().round(), tot[:, 4:].mean().round(), " min/max wave:", tot.min().round(), tot.max().round())
