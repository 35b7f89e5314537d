"""Small batches through the split-K forms of conv3x3_x3.hip (debug switch split_x3_small) against exact fp32."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import get_engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
crops = synth.smooth_crops_u8(n, 3)
e0 = eng.embed_u8(crops)
eng.set_precision(2)
sw = eng.debug_switches_from_env()
for m in (n, 1, 7, 33):
    e2 = eng.embed_u8(crops[:m] if m <= n else synth.smooth_crops_u8(m, 3))
    ref = e0[:m] if m <= n else None
    if ref is None:
        eng.set_precision(0); ref = eng.embed_u8(synth.smooth_crops_u8(m, 3)); eng.set_precision(2)
    print("[%s] %d crops: fp32-class vs exact fp32 %.2e" % (sw, m, float(np.abs(e2 - ref).max() / np.abs(ref).max())))
