"""Swin fp16-storage mode: is an image's embedding independent of the batch it comes in, and are repeated runs identical?
(The 64- and 128-wide instantiations of a linear round their epilogues differently: the tile shape must not depend on the batch.)
    python tools/swin_batch_check.py            REID_DEBUG_SWITCHES=f16_cfg=128323 forces one tile shape"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import get_engine
eng = get_engine(0)
eng.debug_switches_from_env()
sd = synth.swin_state_dict(0) if hasattr(synth, "swin_state_dict") else None
from reid_amd import weights as W
blob, manifest = W.pack_swin(sd)[:2]
eng.load_swin(blob, manifest)
eng.set_precision(1)
rng = np.random.default_rng(0)
x = rng.normal(size=(5, 3, 224, 224)).astype(np.float32)
a1 = eng.swin_embed_f32_nchw(x[:1]); a2 = eng.swin_embed_f32_nchw(x[:1]); b1 = eng.swin_embed_f32_nchw(x); b2 = eng.swin_embed_f32_nchw(x)
e = lambda r: r[0] if isinstance(r, tuple) else r
print("n=1 repeat equal:", np.array_equal(e(a1), e(a2)), " n=5 repeat equal:", np.array_equal(e(b1), e(b2)),
      " n=1 vs n=5[:1] equal:", np.array_equal(e(a1), e(b1)[:1]), " max diff %.3g" % np.abs(e(a1) - e(b1)[:1]).max())
