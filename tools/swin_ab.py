"""Swin in the fp32-class mode, per value of one debug switch: error against the exact-fp32 mode on a few images, batch independence,
and ms per pass of n images (device-resident input):   python tools/swin_ab.py [n=512] [switch=lin_x3] [values=0,1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
name = sys.argv[2] if len(sys.argv) > 2 else "lin_x3"
values = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]
eng = get_engine(0)
eng.load_swin(*weights.pack_swin(synth.swin_state_dict(0))[:2])
rng = np.random.default_rng(0)
small = rng.normal(size=(5, 3, 224, 224)).astype(np.float32)
e = lambda r: r[0] if isinstance(r, tuple) else r
eng.set_precision(0)
ref = e(eng.swin_embed_f32_nchw(small))
eng.set_precision(2)
big = parallel.DevArray.from_numpy(eng, rng.normal(size=(n, 3, 224, 224)).astype(np.float32))
out = parallel.DevArray(eng, (n, 96))
for rep in range(2):
    for v in values:
        eng.debug_switch(name, v)
        got5 = e(eng.swin_embed_f32_nchw(small))
        got1 = e(eng.swin_embed_f32_nchw(small[:1]))
        err = float(np.abs(got5 - ref).max() / np.abs(ref).max())
        for _ in range(2):
            eng.swin_embed_dev(big.ptr, n, 224, 224, out.ptr)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.swin_embed_dev(big.ptr, n, 224, 224, out.ptr)
        eng.sync()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print("%s=%d: vs exact fp32 %.2e; n=1 equals n=5[:1]: %s; %d images %.2f ms = %.0f img/s" % (name, v, err, np.array_equal(got1, got5[:1]), n, ms, n / ms * 1e3), flush=True)
