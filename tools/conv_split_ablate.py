"""Where does the fp32-class LDS-halo convolution spend its time?  One layer shape, the PAIR loop with phases switched off
(reid_debug_conv_split).  python tools/conv_split_ablate.py [crops]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi, synth, weights
from reid_amd._ffi import check
from reid_amd.engine import get_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
fn = _ffi.debug_lib().reid_debug_conv_split
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_float)]
NAMES = {0: "full", 1: "no weight DMA", 2: "no halo DMA", 3: "no DMA", 4: "no MFMA", 8: "no fragment reads", 12: "no reads, no MFMA (barriers + DMA)",
         7: "no DMA, no MFMA (reads + barriers)", 11: "no DMA, no reads (MFMA + barriers)", 16: "no barriers", 15: "barriers only", 31: "nothing", 32: "no epilogue", 47: "barriers only, no epilogue", 64: "no output stores", 79: "barriers only, no stores", 63: "empty kernel", 35: "no DMA, no epilogue", 128: "zero operands", 131: "zero operands, no DMA"}
for name, h, w, c, cout in (("L4 512->512 16x8", 16, 8, 512, 512), ("L3 256->256 16x8", 16, 8, 256, 256), ("L2 128->128 32x16", 32, 16, 128, 128)):
    fl = 2.0 * n * h * w * cout * 9 * c
    for rep in range(2):
        line = []
        for ab in (0, 128, 3, 131):
            ms = C.c_float()
            check(fn(eng.h, n, h, w, c, cout, ab, 5, C.byref(ms)))
            line.append((ab, ms.value))
        base = line[0][1]
        print("%s, %d crops: full %.3f ms = %.0f TF/s algorithmic (%.2f PF on the pipe)" % (name, n, base, fl / base / 1e9, 3 * fl / base / 1e12))
        for ab, ms in line[1:]:
            print("      %-40s %.3f ms  (%.0f %%)" % (NAMES[ab], ms, 100 * ms / base))
