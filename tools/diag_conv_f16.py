"""Stamps of the loader-wave LDS-halo conv kernel: cycles a compute wave spends at the tile barrier vs in reads+MFMAs."""
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
fn = _ffi.debug_lib().reid_debug_conv_f16
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_float)]
dg = _ffi.debug_lib().reid_debug_conv_diag
dg.restype = C.c_int
dg.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for name, h, w, cin, cout in (("L2 128->128 32x16", 32, 16, 128, 128), ("L3 256->256 16x8", 16, 8, 256, 256), ("L4 512->512 16x8", 16, 8, 512, 512),
                              ("L4 as split: 1536->512 16x8", 16, 8, 1536, 512), ("L1 as split: 192->64 64x32", 64, 32, 192, 64)):
    check(dg(eng.h, 1, None))
    ms = C.c_float()
    check(fn(eng.h, n, h, w, cin, cout, 3, 1, 1, 2000001, 1, C.byref(ms)))
    raw = np.zeros(64 * 8 * 5, np.uint64)
    check(dg(eng.h, 0, raw.ctypes.data_as(C.c_void_p)))
    d = raw[:64 * 8 * 4].reshape(64, 8, 4).astype(np.float64)
    epi = raw[64 * 8 * 4:].reshape(64, 8).astype(np.float64)
    nt = d[0, 0, 2]
    if nt == 0:     # three-taps-per-barrier build (Cout = 64 tiles): no stamps in that loop
        print("%s: %.0f TF (kernel %.1f us; no stamps in the three-taps-per-barrier loop)" % (name, 2.0 * n * h * w * cout * 9 * cin / (ms.value * 1e-3) / 1e12, ms.value * 1e3))
        continue
    print("%s: %.0f TF stamped; per tile: barrier %.0f  reads+mfma %.0f cycles (MFMA alone: 512/wave, 1024/SIMD); tiles %d"
          % (name, 2.0 * n * h * w * cout * 9 * cin / (ms.value * 1e-3) / 1e12, d[..., 0].mean() / nt, d[..., 1].mean() / nt, nt))
    loop = (d[..., 0] + d[..., 1]).mean()
    e4 = epi[:, :4].mean(0)
    print("     cycles per block: entry->loop end %.0f (loop itself %.0f, so prologue+stamps %.0f); epilogue: pass1 %.0f | sync+stats %.0f | LDS writes %.0f | sync+stores+drain %.0f; kernel %.1f us"
          % (d[..., 3].mean(), loop, d[..., 3].mean() - loop, e4[0], e4[1], e4[2], e4[3], ms.value * 1e3))
