"""Kernel experiments: times single fp16 implicit-GEMM convolutions (layer shapes of ResNet18-SE) for the tile / ring
configurations compiled into libreid_hip.so (cfg = BN*1000 + BK*10 + NST).  python tools/bench_conv_f16.py [n_crops]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import get_engine
from reid_amd import _ffi
from reid_amd._ffi import check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = get_engine(0)
blob, manifest, _ = weights.pack_seres18(synth.seres18_state_dict(0))
eng.load_seres18(blob, manifest)
fn = _ffi.debug_lib().reid_debug_conv_f16
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_float)]
layers = [("L1 64->64 64x32", 64, 32, 64, 64), ("L2 128->128 32x16", 32, 16, 128, 128),
          ("L3 256->256 16x8", 16, 8, 256, 256), ("L4 512->512 16x8", 16, 8, 512, 512)]
cfgs = [256642, 128642, 64642, 2000000, 2000001]   # 2000000 = LDS-halo kernel (conv3x3_f16.hip)
for name, h, w, cin, cout in layers:
    flops = 2.0 * n * h * w * cout * 9 * cin
    best = {}
    for rep in range(2):
        for cfg in cfgs:
            if cfg < 2000000 and cout % ((cfg % 1000000) // 1000):
                continue
            ms = C.c_float()
            check(fn(eng.h, n, h, w, cin, cout, 3, 1, 1, cfg, 10, C.byref(ms)))
            best[cfg] = min(ms.value, best.get(cfg, 1e9))
    print(name, " ".join("%d:%.0fTF" % (k, flops / (v * 1e-3) / 1e12) for k, v in sorted(best.items(), key=lambda kv: kv[1])))

# dense GEMMs of the same sizes (no im2col gather): separates the gather from the tile loop
gf = _ffi.debug_lib().reid_debug_gemm_f16
gf.restype = C.c_int
gf.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.POINTER(C.c_float), C.c_void_p]
for name, m, nn, k in (("dense L4-size 32768x512x4608", n * 128, 512, 4608), ("dense 8192x8192x4096", 8192, 8192, 4096),
                       ("dense L1-size %dx64x576" % (n * 2048), n * 2048, 64, 576)):
    best = {}
    for rep in range(2):
        for cfg in cfgs:
            if cfg >= 2000000 or nn % ((cfg % 1000000) // 1000):
                continue
            ms = C.c_float()
            check(gf(eng.h, m, nn, k, cfg, 5, C.byref(ms), None))
            best[cfg] = min(ms.value, best.get(cfg, 1e9))
    print(name, " ".join("%d:%.0fTF" % (c, 2.0 * m * nn * k / (v * 1e-3) / 1e12) for c, v in sorted(best.items(), key=lambda kv: kv[1])))
