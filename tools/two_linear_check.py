"""The fused pair of linears (csrc/two_linear_f16.hip) against (a) float64, (b) the two launches it replaces (gemm_f16.hip linear
builds through reid_debug_linear_rows), (c) itself on copies of the same rows at other positions (bit-identical), and its time.
python tools/two_linear_check.py [tokens] [ablate bits: 1 = no weight refills, 2 = no barriers - timing only, wrong results]"""
import ctypes as C
import math
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi, synth, weights
from reid_amd._ffi import check
from reid_amd.engine import get_engine

eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
eng.set_precision(2)
dbg = _ffi.debug_lib()
two = dbg.reid_debug_two_linear
two.restype = C.c_int
two.argtypes = [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p] * 4
lin = dbg.reid_debug_linear_rows
lin.restype = C.c_int
lin.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]
rng = np.random.default_rng(0)
erf = np.vectorize(math.erf)


def case(m, c, hid, act, iters=1, ln=False):
    R = 61
    base = rng.normal(size=(R, c)).astype(np.float32)
    ids = np.asarray([(i * 5 + i // 7) % R for i in range(m)])
    x = np.ascontiguousarray(base[ids])
    rb = rng.normal(size=(R, c)).astype(np.float32)
    res = np.ascontiguousarray(rb[ids])
    w1 = (rng.normal(size=(hid, c)) / np.sqrt(c)).astype(np.float32)
    b1 = rng.normal(size=hid).astype(np.float32)
    w2 = (rng.normal(size=(c, hid)) / np.sqrt(hid)).astype(np.float32)
    b2 = rng.normal(size=c).astype(np.float32)
    out = np.empty((m, c), np.float32)
    ms = C.c_float(0)
    g = (1.0 + 0.1 * rng.normal(size=c)).astype(np.float32)
    bt = (0.1 * rng.normal(size=c)).astype(np.float32)
    if ln:      # rows with a mean and a spread of their own
        base = (base * rng.uniform(0.5, 3.0, size=(R, 1)) + rng.normal(size=(R, 1))).astype(np.float32)
        x = np.ascontiguousarray(base[ids])
    check(two(eng.h, x.ctypes.data, w1.ctypes.data, b1.ctypes.data, w2.ctypes.data, b2.ctypes.data, res.ctypes.data, m, c, hid, act, iters,
              out.ctypes.data, C.addressof(ms), g.ctypes.data if ln else None, bt.ctypes.data if ln else None))
    # (a) float64
    b64 = base.astype(np.float64)
    if ln:
        b64 = (b64 - b64.mean(1, keepdims=True)) / np.sqrt(b64.var(1, keepdims=True) + 1e-5) * g + bt
        x = np.ascontiguousarray(b64.astype(np.float32)[ids])      # what the two-launch form is given
    h = b64 @ w1.T.astype(np.float64) + b1
    if act:
        h = 0.5 * h * (1.0 + erf(h / math.sqrt(2.0)))
    ref = (h @ w2.T.astype(np.float64) + b2 + rb)[ids]
    err = np.abs(out - ref).max() / np.abs(ref).max()
    # (b) the two launches
    hid_out = np.empty((m, hid), np.float32)
    check(lin(eng.h, x.ctypes.data, w1.ctypes.data, b1.ctypes.data, None, m, hid, c, 2, 2 | act, hid_out.ctypes.data))
    un = np.empty((m, c), np.float32)
    check(lin(eng.h, hid_out.ctypes.data, w2.ctypes.data, b2.ctypes.data, res.ctypes.data, m, c, hid, 2, 0, un.ctypes.data))
    err_un = np.abs(un - ref).max() / np.abs(ref).max()
    dfu = np.abs(out - un).max() / np.abs(ref).max()
    # (c) copies
    first = [int(np.flatnonzero(ids == r)[0]) for r in range(R)]
    ndiff = int((out != out[first][ids]).sum())
    flops = 4.0 * m * c * hid * 3
    print("m %7d c %3d hid %4d act %d ln %d: fused vs f64 %.2e, two launches vs f64 %.2e, fused vs two launches %.2e, %d elements differ between "
          "copies%s" % (m, c, hid, act, ln, err, err_un, dfu, ndiff,
                        "; %.1f us per launch = %.0f TFLOP/s (f16 products)" % (ms.value * 1e3, flops / ms.value / 1e9) if iters > 1 else ""))
    return err, ndiff


bad = 0
for (m, c, hid, act) in ((2048, 96, 384, 1), (2048 + 32, 96, 96, 0), (4096 + 160, 96, 384, 1), (1024 + 49, 96, 96, 0), (1568 * 3, 96, 384, 1),
                         (2048, 192, 768, 1), (1024 + 16 * 49, 192, 192, 0)):
    for ln in (False, True):
        e, nd = case(m, c, hid, act, ln=ln)
        bad += (e > 3e-6) + nd
big = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 3136
if len(sys.argv) > 2:
    print("ablation bits %s: the errors below are expected" % sys.argv[2])
    check(dbg.reid_debug_two_linear_ablate(eng.h, int(sys.argv[2])))
case(big, 96, 384, 1, iters=10)
case(big, 96, 384, 1, iters=10, ln=True)
case(big, 96, 96, 0, iters=10)
case(big, 96, 384, 0, iters=10)
if True:
    case(big // 4, 192, 768, 1, iters=10)
    case(big // 4, 192, 768, 1, iters=10, ln=True)
    case(big // 4, 192, 192, 0, iters=10)
print("FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
