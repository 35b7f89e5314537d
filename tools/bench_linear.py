"""Every Linear shape of Swin-T (256 images per pass) through the exact-fp32 and the fp16-storage GEMM, with the time the
layer's HBM traffic alone would take beside it.  python tools/bench_linear.py [f32|f16|both] [images]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi, synth, weights
from reid_amd._ffi import check
from reid_amd.engine import get_engine

which = sys.argv[1] if len(sys.argv) > 1 else "both"
imgs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
fn = _ffi.debug_lib().reid_debug_linear
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
tot = {0: 0.0, 1: 0.0}
fl_tot = 0.0
for stage, (tok, blocks) in enumerate(((3136, 2), (784, 2), (196, 6), (49, 2))):
    c = 96 << stage
    m = imgs * tok
    for name, n, k, epi in (("qkv", 3 * c, c, 0), ("proj", c, c, 2), ("fc1", 4 * c, c, 1), ("fc2", c, 4 * c, 2)):
        flops = 2.0 * m * n * k
        line = "stage %d %-4s M=%7d N=%4d K=%4d x%d" % (stage + 1, name, m, n, k, blocks)
        for h in (0, 1):
            if which not in ("both", "f16" if h else "f32"):
                continue
            es = 2 if h else 4
            byt = m * k * es + n * k * es + (m * n * 8 if epi == 2 else m * n * es)
            ms = C.c_float()
            check(fn(eng.h, m, n, k, h | (epi << 1), 10, C.byref(ms)))
            tot[h] += ms.value * blocks
            line += "   %s %7.3f ms %6.1f TF/s (HBM at 4.5 TB/s: %6.3f ms)" % ("f16" if h else "f32", ms.value, flops / ms.value / 1e9, byt / 4.5e9)
        fl_tot += flops * blocks
        print(line, flush=True)
print("all linears of one pass of %d images: " % imgs + "  ".join("%s %.2f ms = %.1f TF/s" % ("f16" if h else "f32", tot[h], fl_tot / tot[h] / 1e9) for h in (0, 1) if tot[h]))
