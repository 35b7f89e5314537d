"""Where a K-tile of conv_f32.hip spends its cycles: s_memtime stamps of the diagnostic build (shares, not lengths).
Segments per wave and K-tile of the LDS-DMA kernel: wait for the wave's own DMA pieces, block barrier, issue of the next tile's
DMA, fragment reads + MFMAs.   python tools/diag_conv_f32.py [n_crops]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi, synth, weights
from reid_amd.engine import get_engine
from reid_amd._ffi import check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = get_engine(0)
blob, manifest, _ = weights.pack_seres18(synth.seres18_state_dict(0))
eng.load_seres18(blob, manifest)
dbg = _ffi.debug_lib()
fn = dbg.reid_debug_conv_f32
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 11 + [C.POINTER(C.c_float)]
dg = dbg.reid_debug_conv_diag
dg.restype = C.c_int
dg.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
layers = [("L1 conv2 64->64", 64, 32, 64, 64, 3, 1, 1, 15), ("L2 conv 128->128", 32, 16, 128, 128, 3, 1, 1, 15),
          ("L4 conv1 256->512", 16, 8, 256, 512, 3, 1, 1, 8), ("L4 conv 512->512", 16, 8, 512, 512, 3, 1, 1, 15)]
for name, h, w, cin, cout, r, stride, pad, flags in layers:
    nk = r * r * cin // 32
    ms = C.c_float()
    check(dg(eng.h, 1, None))
    check(fn(eng.h, n, h, w, cin, cout, r, stride, pad, (flags & ~1) | 2, 1, 1, C.byref(ms)))
    out = np.zeros(64 * 8 * 5, np.uint64)
    check(dg(eng.h, 0, out.ctypes.data_as(C.c_void_p)))
    d = out.reshape(64, 8, 5)[:, :4, :].astype(np.float64) / nk     # cycles per K-tile
    med = np.median(d.reshape(-1, 5), 0)
    mf = 64 * (2 if cout >= 128 else 1) * 64 // 2
    print("%-20s LDS-DMA kernel, cycles per K-tile (median over 256 waves): wait own DMA %.0f  barrier %.0f  issue next DMA %.0f  "
          "fragment reads + %d MFMAs %.0f  | sum %.0f, matrix-pipe time of the SIMD's two waves %d" % (name, med[0], med[1], med[2], mf // 32, med[3], med[:4].sum(), mf * 2))
