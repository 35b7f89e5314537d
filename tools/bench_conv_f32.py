"""Kernel experiments: times the exact-fp32 convolutions of ResNet18-SE (every conv shape of the network, with the fusions
it runs with) for the round-1 kernel (variant 0, gemm_f32_kernel<A_IM2COL>), the register-staged conv_f32.hip kernel (variant 2) and its
LDS-DMA kernel (variant 1, what the forward runs), interleaved in one process.  python tools/bench_conv_f32.py [n_crops]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi, synth, weights
from reid_amd.engine import get_engine
from reid_amd._ffi import check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 2, 1]   # 0 round-1 kernel, 2 conv_f32 without the priority switch, 1 conv_f32
eng = get_engine(0)
blob, manifest, _ = weights.pack_seres18(synth.seres18_state_dict(0))
eng.load_seres18(blob, manifest)
fn = _ffi.debug_lib().reid_debug_conv_f32
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] + [C.c_int] * 11 + [C.POINTER(C.c_float)]
# name, h, w, cin, cout, r, stride, pad, flags (1 affine-in, 2 BN epilogue, 4 residual+ReLU, 8 stats), launches per forward
layers = [
    ("L1 conv1 64->64 64x32 +stats", 64, 32, 64, 64, 3, 1, 1, 8, 2),
    ("L1 conv2 64->64 +aff+bn+res+stats", 64, 32, 64, 64, 3, 1, 1, 15, 2),
    ("L2 conv1 64->128 s2 +stats", 64, 32, 64, 128, 3, 2, 1, 8, 1),
    ("L2 conv 128->128 32x16 +aff+bn+res+stats", 32, 16, 128, 128, 3, 1, 1, 15, 3),
    ("L2 ds 1x1 64->128 s2 +bn", 64, 32, 64, 128, 1, 2, 0, 2, 1),
    ("L3 conv1 128->256 s2 +stats", 32, 16, 128, 256, 3, 2, 1, 8, 1),
    ("L3 conv 256->256 16x8 +aff+bn+res+stats", 16, 8, 256, 256, 3, 1, 1, 15, 3),
    ("L3 ds 1x1 128->256 s2 +bn", 32, 16, 128, 256, 1, 2, 0, 2, 1),
    ("L4 conv1 256->512 16x8 +stats", 16, 8, 256, 512, 3, 1, 1, 8, 1),
    ("L4 conv 512->512 16x8 +aff+bn+res+stats", 16, 8, 512, 512, 3, 1, 1, 15, 3),
    ("L4 ds 1x1 256->512 +bn", 16, 8, 256, 512, 1, 1, 0, 2, 1),
]
tot = {v: 0.0 for v in variants}
tot_fl = 0.0
for name, h, w, cin, cout, r, stride, pad, flags, cnt in layers:
    ho, wo = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - r) // stride + 1
    flops = 2.0 * n * ho * wo * cout * r * r * cin
    best = {}
    for rep in range(3):
        for v in variants:
            ms = C.c_float()
            # variant 1 = the LDS-DMA flow: no affine in the loader (flag 1 off), the BatchNorm half in conv1's epilogue (flag 2 on)
            fl = ((flags & ~1) | 2) if v == 1 else flags
            check(fn(eng.h, n, h, w, cin, cout, r, stride, pad, fl, v, 5, C.byref(ms)))
            best[v] = min(ms.value, best.get(v, 1e9))
    for v in variants:
        tot[v] += best[v] * cnt
    tot_fl += flops * cnt
    print("%-44s" % name, "  ".join("v%d: %7.3f ms %6.1f TF" % (v, best[v], flops / (best[v] * 1e-3) / 1e12) for v in variants), flush=True)
print("all conv launches of one forward of %d crops:" % n,
      "  ".join("v%d: %.2f ms = %.1f TF/s (%.3f of 157.3)" % (v, tot[v], tot_fl / (tot[v] * 1e-3) / 1e12, tot_fl / (tot[v] * 1e-3) / 1e12 / 157.3) for v in variants))
