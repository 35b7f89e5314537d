"""sha256 of Swin-T embeddings of 96 synthetic images in the three precisions (A/B of two builds: REID_HIP_LIB=... python tools/probes/swin_hash.py)"""
import hashlib
import sys

import numpy as np

sys.path.insert(0, ".")
from reid_amd import synth, weights
from reid_amd.engine import Engine

eng = Engine(0)
sd = synth.swin_state_dict(0)
eng.load_swin(*weights.pack_swin(sd)[:2])
x = synth.images_f32(96, 5)
for mode in (0, 1, 2):
    eng.set_precision(mode)
    for chunk in (32, 96):
        eng.set_chunk(chunk)
        e = eng.swin_embed_f32_nchw(x)
        print(mode, chunk, hashlib.sha256(np.ascontiguousarray(e).tobytes()).hexdigest()[:16], float(np.abs(e).max()))
