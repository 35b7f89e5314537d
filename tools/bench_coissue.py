"""What a staging wave can issue beside the other wave's back-to-back v_mfma_f32_32x32x2_f32 on its SIMD (microbench.hip).
python tools/bench_coissue.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi
from reid_amd.engine import get_engine
from reid_amd._ffi import check

eng = get_engine(0)
fn = _ffi.debug_lib().reid_debug_coissue
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
iters = 32
names = {0: "v_fma_f32", 1: "ds_write_b128", 2: "global_load_dwordx4", 3: "v_fma_f32 under s_setprio 3"}
for mode in (0, 3, 1, 2):
    res = {}
    for roles in (1, 2, 3):
        a, b = C.c_double(), C.c_double()
        check(fn(eng.h, mode, iters, roles, C.byref(a), C.byref(b)))
        res[roles] = (a.value, b.value)
    nm, no = iters * 16, iters * 64
    print("%-28s MFMA wave alone %.1f cyc/MFMA | other wave alone %.1f cyc/op | together: %.1f cyc/MFMA, %.1f cyc/op"
          % (names[mode], res[1][0] / nm, res[2][1] / no, res[3][0] / nm, res[3][1] / no))
    a, b = C.c_double(), C.c_double()
    check(fn(eng.h, mode, iters, 5, C.byref(a), C.byref(b)))
    a1 = a.value
    check(fn(eng.h, mode, iters, 7, C.byref(a), C.byref(b)))
    print("%-28s beside v_mfma_f32_32x32x16_f16: MFMA wave alone %.1f cyc/MFMA | together: %.1f cyc/MFMA, %.1f cyc/op"
          % ("", a1 / nm, a.value / nm, b.value / no))
