"""MFMA-shape experiment: the halo kernel's inner loop (LDS fragment reads + MFMAs of a 64x64 wave tile) with the 32x32x16
and the 16x16x32 f16 MFMA on random operands, several ms per launch so that the clock settles.  python tools/bench_mfma_shape.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd.engine import get_engine
from reid_amd import _ffi
from reid_amd._ffi import check

eng = get_engine(0)
fn = _ffi.debug_lib().reid_debug_mfma_shape
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
for blocks in (256, 512):
    for rep in range(3):
        out = []
        for shape in (32, 16):
            tf = C.c_float()
            check(fn(eng.h, shape, 40000, blocks, C.byref(tf)))
            out.append("%dx%d: %.0f TF" % (shape, shape, tf.value))
        print("blocks", blocks, " ".join(out))
