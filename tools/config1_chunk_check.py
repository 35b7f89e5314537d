"""The config-1 parity sets (tests/golden/config1.npz: 256 crops, arg-min of the (1 - cos) / 2 matrix against the reference's) under
different pass sizes: rows whose arg-min differs from the reference's, per precision.  python tools/config1_chunk_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi, synth, weights
from reid_amd.engine import get_engine

eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config1.npz"))
for chunk in (64, 128, 256, 1024):
    eng.set_chunk(chunk)
    for precision in (0, 2):
        eng.set_precision(precision)
        for tag, fn, seed in (("rand0", synth.crops_u8, 0), ("smooth5", synth.smooth_crops_u8, 5)):
            emb = eng.embed_u8(fn(256, seed))
            d = eng.distmat(emb, emb, _ffi.METRIC_COS_HALF).copy()
            np.fill_diagonal(d, np.inf)
            flips = np.flatnonzero(d.argmin(1) != g[tag + "_argmin"])
            print("pass size %4d precision %d %-7s: %d arg-mins differ from the reference's%s" % (
                chunk, precision, tag, len(flips), "" if not len(flips) else " (reference top-2 gaps %s)" % g[tag + "_gap"][flips]))
eng.set_precision(0)
