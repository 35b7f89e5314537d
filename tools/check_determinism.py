"""Run the fp16 embed path several times on the same crops and report whether the results are bit-identical
(a difference means a race in a kernel).  python tools/check_determinism.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import get_engine

eng = get_engine(0)
sd = synth.seres18_state_dict(0)
blob, manifest, _ = weights.pack_seres18(sd)
eng.load_seres18(blob, manifest)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
crops = synth.smooth_crops_u8(n, 3)
for prec in (1, 0):
    eng.set_precision(prec)
    outs = [eng.embed_u8(crops) for _ in range(4)]
    for keep in (1, 2):
        eng.debug_keep(keep)
        outs.append(eng.embed_u8(crops))
        stages = [eng.debug_stage(s, n) for s in range(1, 11)]
        eng.debug_keep(0)
        if keep == 1:
            st1 = stages
        else:
            print("prec", prec, "stage max|keep1-keep2|:", [float(np.abs(a - b).max()) for a, b in zip(st1, stages)])
    print("prec", prec, "max |run_i - run_0|:", [float(np.abs(o - outs[0]).max()) for o in outs], "(last two: debug_keep 1, 2)")
