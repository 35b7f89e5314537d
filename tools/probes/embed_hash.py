"""sha256 of ResNet18-SE embeddings at several pass sizes and precisions (A/B of two builds: REID_HIP_LIB=... python tools/probes/embed_hash.py)"""
import hashlib
import sys

import numpy as np

sys.path.insert(0, ".")
from reid_amd import synth, weights
from reid_amd.engine import Engine

eng = Engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
crops = synth.crops_u8(300, 1)
sw = eng.debug_switches_from_env()
for mode in (0, 2):
    eng.set_precision(mode)
    for n in (1, 7, 20, 30, 33, 47, 64, 100, 130, 150, 190, 200, 255, 256, 260, 300):
        e = eng.embed_u8(crops[:n])
        print(mode, n, hashlib.sha256(np.ascontiguousarray(e).tobytes()).hexdigest()[:16])
