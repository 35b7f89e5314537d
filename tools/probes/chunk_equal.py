"""4096 crops in mode 2 with passes of 1024 (the unrolled kernels) and one pass of 4096 (layer 1's input passes 2 GB: the looped conv3x3_x3m16_kernel): same embeddings?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine
eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
n = 4096
crops = parallel.DevArray.from_numpy(eng, synth.crops_u8(n, 1))
emb = parallel.DevArray(eng, (n, 512))
eng.set_precision(2)
res = {}
for chunk in (1024, 4096):
    eng.set_chunk(chunk)
    eng.embed_u8_dev(crops.ptr, n, emb.ptr)
    res[chunk] = emb.numpy().copy()
a, b = res[1024], res[4096]
print("finite:", np.isfinite(a).all(), np.isfinite(b).all(), " max |diff| / max |emb| = %.3g" % (np.abs(a - b).max() / np.abs(a).max()), " identical rows: %d of %d" % ((a == b).all(1).sum(), n))
for sw in (0,):
    eng.debug_switch("x3_unroll", sw)
    eng.set_chunk(1024)
    eng.embed_u8_dev(crops.ptr, n, emb.ptr)
    c = emb.numpy()
    print("x3_unroll=%d (looped kernels) vs unrolled: identical rows %d of %d" % (sw, (a == c).all(1).sum(), n))
eng.debug_switch("x3_unroll", 3)
