"""Plain dense f16 GEMMs of given shapes under every tile / ring configuration of gemm_f16.hip (main-loop comparison; the Swin
linears' K = 3 C virtual columns).  python tools/gemm_cfg_sweep.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights, _ffi
from reid_amd._ffi import check
from reid_amd.engine import get_engine

eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
gf = _ffi.debug_lib().reid_debug_gemm_f16
gf.restype = C.c_int
gf.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.POINTER(C.c_float), C.c_void_p]
shapes = ((200704, 1152, 1152, "stage 3 qkv"), (200704, 384, 1152, "stage 3 out / post"), (200704, 1536, 1152, "stage 3 fc1"),
          (200704, 384, 4608, "stage 3 fc2"), (50176, 2304, 2304, "stage 4 qkv"), (50176, 768, 9216, "stage 4 fc2"))
cfgs = (128323, 128324, 128643, 128642, 256642, 256324, 64323, 64643, 1128323, 1128643, 1256324)
for m, n, k, name in shapes:
    line = "%-18s M=%6d N=%4d K=%4d:" % (name, m, n, k)
    for cfg in cfgs:
        if n % ((cfg % 1000000) // 1000):
            continue
        ms = C.c_float()
        check(gf(eng.h, m, n, k, cfg, 5, C.byref(ms), None))
        line += "  %d %.0f" % (cfg, 2.0 * m * n * k / ms.value / 1e9)
    print(line + "   (TF/s)", flush=True)
