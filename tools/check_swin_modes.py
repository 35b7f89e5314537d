"""Swin-T, 256 images of 224x224 per pass in the three arithmetic modes (0 exact fp32, 2 fp32-class, 1 fp16 storage): time per pass and
the difference of the embeddings from mode 0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine
eng = get_engine(0)
eng.load_swin(*weights.pack_swin(synth.swin_state_dict(0))[:2])
x = synth.images_f32(256, 2)
dx = parallel.DevArray.from_numpy(eng, x)
emb = parallel.DevArray(eng, (256, 96))
eng.set_chunk(256)
ref = None
for mode in (0, 2, 1):
    eng.set_precision(mode)
    for _ in range(2):
        eng.swin_embed_dev(dx.ptr, 256, 224, 224, emb.ptr)
    eng.timer_start()
    for _ in range(3):
        eng.swin_embed_dev(dx.ptr, 256, 224, 224, emb.ptr)
    ms = eng.timer_stop() / 3
    e = emb.numpy()
    if ref is None: ref = e
    cos = (e * ref).sum(1) / np.linalg.norm(e, axis=1) / np.linalg.norm(ref, axis=1)
    print("swin mode %d: 256 images in %.2f ms = %.1f k img/s; vs mode 0: max rel %.2e, 1-cos %.1e" % (mode, ms, 256 / ms, np.abs(e - ref).max() / np.abs(ref).max(), (1 - cos).max()))
