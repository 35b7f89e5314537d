"""precision 2 ("fp32-class": hi/lo-split operands on the f16 matrix pipe) against the reference fixture and the exact-fp32 mode:
stage taps, embedding cosine, config-1 arg-min flips, and the time of a 1024-crop pass in both modes.  python tools/check_split_mode.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import seres18
from reid_amd import _ffi, parallel, synth, weights
from reid_amd.engine import get_engine

eng = get_engine(0)
for tag, crops_fn in (("seed0", synth.crops_u8), ("smooth1", synth.smooth_crops_u8)):
    g = np.load(os.path.join(ROOT, "tests", "golden", "seres18_%s.npz" % tag))
    seed, n = int(g["seed"]), int(g["n"])
    sd = synth.seres18_state_dict(seed)
    eng.load_seres18(*weights.pack_seres18(sd)[:2])
    crops = crops_fn(n, seed)
    taps = {}
    seres18.forward(sd, seres18.preprocess_u8(crops), taps)
    names = ["stem", "pool0"] + [b[0] for b in synth.SERES18_BLOCKS] + ["gem"]
    for mode in (0, 2):
        eng.set_precision(mode)
        eng.debug_keep(True)
        emb = eng.embed_u8(crops)
        worst = 0.0
        for s, name in enumerate(names):
            t = taps[name]
            want = t.permute(0, 2, 3, 1).contiguous().numpy().reshape(-1) if t.dim() == 4 else t.numpy().reshape(-1)
            got = eng.debug_stage(s, n)
            worst = max(worst, float(np.abs(got - want).max() / max(1e-6, np.abs(want).max())))
        eng.debug_keep(False)
        cos = (emb * g["emb"]).sum(1) / np.linalg.norm(emb, axis=1) / np.linalg.norm(g["emb"], axis=1)
        print("%s mode %d: stage taps max rel err %.2e (limit 2e-5), emb rel err %.2e, 1 - cos %.1e" %
              (tag, mode, worst, np.abs(emb - g["emb"]).max() / np.abs(g["emb"]).max(), (1 - cos).max()))
g1 = np.load(os.path.join(ROOT, "tests", "golden", "config1.npz"))
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
for tag, crops in (("rand0", synth.crops_u8(256, 0)), ("smooth5", synth.smooth_crops_u8(256, 5))):
    for mode in (0, 2):
        eng.set_precision(mode)
        emb = eng.embed_u8(crops)
        d = eng.distmat(emb, emb, _ffi.METRIC_COS_HALF)
        np.fill_diagonal(d, np.inf)
        flips = np.flatnonzero(d.argmin(1) != g1[tag + "_argmin"])
        print("config1 %s mode %d: arg-min flips %d of 256%s" % (tag, mode, len(flips), (" (reference top-2 gaps <= %.1e)" % g1[tag + "_gap"][flips].max()) if len(flips) else ""))
crops = parallel.DevArray.from_numpy(eng, synth.crops_u8(1024, 1))
emb = parallel.DevArray(eng, (1024, 512))
eng.set_chunk(1024)
for mode in (0, 2, 1):
    eng.set_precision(mode)
    for _ in range(2):
        eng.embed_u8_dev(crops.ptr, 1024, emb.ptr)
    eng.timer_start()
    for _ in range(3):
        eng.embed_u8_dev(crops.ptr, 1024, emb.ptr)
    ms = eng.timer_stop() / 3
    print("mode %d: 1024 crops in %.2f ms = %.1f k crops/s" % (mode, ms, 1024 / ms))
eng.set_precision(0)
