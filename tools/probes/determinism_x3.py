"""Mode 2 run-to-run determinism at the pass sizes where split-K / gather kernels meet (reduce-scatter sums in split order: any difference
between two runs of the same pass is a race): python tools/probes/determinism_x3.py"""
import sys

import numpy as np

sys.path.insert(0, ".")
from reid_amd import synth, weights
from reid_amd.engine import get_engine

eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
crops = synth.smooth_crops_u8(520, 3)
eng.set_precision(2)
bad = 0
for n in (1, 7, 21, 30, 33, 48, 62, 66, 96, 120, 130, 190, 256, 300, 520):
    outs = [eng.embed_u8(crops[:n]) for _ in range(6)]
    d = max(float(np.abs(o - outs[0]).max()) for o in outs[1:])
    bad += d != 0.0
    print("n=%4d  max |run_i - run_0| over 5 repeats: %g" % (n, d))
print("fault bits", eng.fault_bits(), "sizes with a difference:", bad)
