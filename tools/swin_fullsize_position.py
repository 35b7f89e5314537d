"""BASELINE configs[2] arrangement (256 distinct images x 16, shuffled, passes of 256): which copies differ from their first copy,
in which pass and at which position.  python tools/swin_fullsize_position.py [precision] [copies]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import get_engine

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 16
eng = get_engine(0)
eng.load_swin(*weights.pack_swin(synth.swin_state_dict(0))[:2])
rng = np.random.default_rng(12)
base = synth.images_f32(256, 2)
ids = np.repeat(np.arange(256), copies)
rng.shuffle(ids)
x = base[ids]
eng.set_precision(prec)
eng.set_chunk(256)
emb = eng.swin_embed_f32_nchw(x)
emb2 = eng.swin_embed_f32_nchw(x)
print("run-to-run identical:", bool(np.array_equal(emb, emb2)))
first = np.asarray([np.flatnonzero(ids == c)[0] for c in range(256)])
ref = emb[first][ids]
bad = np.flatnonzero((emb != ref).any(1))
print("precision %d: %d of %d rows differ from their first copy; max rel %.3e" % (prec, len(bad), len(ids), float(np.abs(emb - ref).max() / np.abs(emb).max())))
print("passes of the differing rows:", np.bincount(bad // 256, minlength=copies).tolist())
print("passes of their first copies:", np.bincount(first[ids[bad]] // 256, minlength=copies).tolist())
# group the copies of each image by value: how many distinct embeddings per image?
ndist = [len({emb[i].tobytes() for i in np.flatnonzero(ids == c)}) for c in range(256)]
print("distinct embeddings per image: histogram", np.bincount(ndist).tolist())
for c in [c for c in range(256) if ndist[c] > 1][:6]:
    rows = np.flatnonzero(ids == c)
    groups = {}
    for r in rows:
        groups.setdefault(emb[r].tobytes(), []).append(int(r))
    print(" image %d:" % c, [[(r // 256, r % 256) for r in g] for g in groups.values()])
# the same images alone, one pass each at position 0: the "canonical" value
alone = eng.swin_embed_f32_nchw(base[:8])
for c in range(8):
    rows = np.flatnonzero(ids == c)
    print(" image %d alone-in-a-pass-of-8 equals copies:" % c, [bool(np.array_equal(alone[c], emb[r])) for r in rows])
