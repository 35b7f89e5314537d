"""Row-position invariance of the f16 linear build (gemm_f16.hip, LIN): identical input rows must give bit-identical output rows
wherever they sit in a 256-row tile.  python tools/linear_row_check.py"""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi, synth, weights
from reid_amd._ffi import check
from reid_amd.engine import get_engine

eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
fn = _ffi.debug_lib().reid_debug_linear_rows
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]
rng = np.random.default_rng(0)


def rows_check(m, n, k, mode, flags, with_res):
    R = 8
    base = rng.normal(size=(R, k)).astype(np.float32)
    ids = np.asarray([(i * 5 + i // 7) % R for i in range(m)])
    x = np.ascontiguousarray(base[ids])
    w = (rng.normal(size=(n, k)) / np.sqrt(k)).astype(np.float32)
    bias = rng.normal(size=n).astype(np.float32)
    res = None
    if with_res:
        rb = rng.normal(size=(R, n)).astype(np.float32)
        res = np.ascontiguousarray(rb[ids])
    out = np.empty((m, n), np.float32)
    check(fn(eng.h, x.ctypes.data, w.ctypes.data, bias.ctypes.data, res.ctypes.data if res is not None else None, m, n, k, mode, flags, out.ctypes.data))
    first = [int(np.flatnonzero(ids == r)[0]) for r in range(R)]
    ref = out[first][ids]
    d = np.argwhere(out != ref)
    print("m %5d n %4d k %4d mode %d flags %d res %d: %d of %d elements differ between copies" % (m, n, k, mode, flags, with_res, len(d), out.size))
    for (i, j) in d[:6]:
        print("    row %d (tile row %d) col %d: %r vs row %d (tile row %d): %r" % (i, i % 256, j, out[i, j], first[ids[i]], first[ids[i]] % 256, ref[i, j]))
    return len(d)


tot = 0
for mode in (1, 2):
    for (n, k) in ((384, 96), (96, 384), (288, 96), (768, 192), (3072, 768), (768, 3072)):
        for flags, res in ((3, 0), (2, 0), (0, 1), (1, 0)):
            tot += rows_check(4096 + 128, n, k, mode, flags, res)
print("TOTAL differing:", tot)
