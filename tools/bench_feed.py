"""Operand-feed ceilings: bytes/s a CU can pull by LDS-DMA vs register loads, by footprint (L2 / Infinity Cache / HBM)
and row size (64-B, 128-B, 1-KiB contiguous pieces)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd.engine import get_engine
from reid_amd import _ffi
from reid_amd._ffi import check

eng = get_engine(0)
fn = _ffi.debug_lib().reid_debug_feed
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
for fp_name, fp in (("2MB(L2)", 2 << 20), ("24MB(L2 agg)", 24 << 20), ("128MB(MALL)", 128 << 20), ("2GB(HBM)", 2 << 30)):
    for rowb, stride in ((64, 64), (64, 1024), (128, 128), (128, 1024), (1024, 1024)):
        out = []
        for mode, infl in ((0, 0), (0, 8), (1, 0)):
            g, t = C.c_float(), C.c_float()
            check(fn(eng.h, mode, fp, rowb, stride, 64, infl, C.byref(g), C.byref(t)))
            out.append("%s:%.0fGB/s/CU(%.1fTB/s)" % (("dma", "dma-pipelined", "regs")[mode if mode else (1 if infl else 0)] if mode == 0 else "regs", g.value, t.value))
        print("%-14s row %4dB stride %5d  %s" % (fp_name, rowb, stride, "  ".join(out)))
