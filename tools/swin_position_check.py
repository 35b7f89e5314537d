"""Swin: is an image's result independent of its POSITION inside a pass?  Copies of two images at even / odd positions of a pass;
prints, per stage (reid_debug_swin_stage), the largest difference between copies of the same image.  python tools/swin_position_check.py [n] [precision]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import get_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = get_engine(0)
eng.load_swin(*weights.pack_swin(synth.swin_state_dict(0))[:2])
base = synth.images_f32(3, 2)
ids = np.asarray([(i * 7 + i // 3) % 3 for i in range(n)])
x = base[ids]
eng.set_precision(prec)
eng.set_chunk(256)
eng.debug_keep(True)
emb = eng.swin_embed_f32_nchw(x)
for st in (range(6) if n <= 256 else []):
    t = eng.debug_swin_stage(st, n)
    worst = 0.0
    for c in range(3):
        rows = np.flatnonzero(ids == c)
        for r in rows[1:]:
            worst = max(worst, float(np.abs(t[r] - t[rows[0]]).max() / np.abs(t[rows[0]]).max()))
    print("stage %d: max rel diff between copies %.3e" % (st, worst))
first = [np.flatnonzero(ids == c)[0] for c in range(3)]
print("emb: bit-equal copies:", bool(np.array_equal(emb, emb[first][ids])), " max rel", float(np.abs(emb - emb[first][ids]).max() / np.abs(emb).max()))
