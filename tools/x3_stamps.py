"""s_memtime stamps of the single-halo unrolled convolution (conv3x3_x3u_kernel<.., 64, 3>) on layer-1 / layer-2 shapes (Gemm16Params.diag: reid_debug_conv_diag)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights, _ffi
from reid_amd.engine import get_engine
from reid_amd._ffi import check

eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
eng.debug_switches_from_env()
fn = _ffi.debug_lib().reid_debug_conv_split
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_float)]
dg = _ffi.debug_lib().reid_debug_conv_diag
dg.restype = C.c_int
dg.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
abl = int(sys.argv[2]) if len(sys.argv) > 2 else 0    # Gemm16Params.ablate: 64 = no fp32 stores, 32 = no epilogue at all
for name, h, w, c, cout in (("L1 64->64 64x32", 64, 32, 64, 64), ("L2 128->128 32x16", 32, 16, 128, 128)):
    ms = C.c_float()
    check(fn(eng.h, n, h, w, c, cout, abl, 10, C.byref(ms)))
    t_plain = ms.value
    check(dg(eng.h, 1, None))
    check(fn(eng.h, n, h, w, c, cout, abl, 1, C.byref(ms)))
    raw = np.zeros(64 * 8 * 5, np.uint64)
    check(dg(eng.h, 0, raw.ctypes.data_as(C.c_void_p)))
    r = raw.reshape(64, 8, 5)[:, :4, :].astype(np.float64)
    m = r.reshape(-1, 5).mean(0)
    print("[ablate %d] %s x %d: %.1f us per launch (%.0f TF); per block (s_memtime units, mean of 64 blocks x 4 waves): set-up %.0f | first halo wait %.0f | later halo waits %.0f | entry->loop end %.0f | epilogue+drain %.0f"
          % (abl, name, n, t_plain * 1e3, 2.0 * n * h * w * cout * 27 * c / (t_plain * 1e-3) / 1e12, m[0], m[1], m[2], m[3], m[4]), flush=True)
