"""conv_x3s_kernel (strided 3x3 / 1x1 convolutions, fp32-class mode) against the forms it replaces and against exact fp32:
python tools/probes/x3s_check.py   (switch conv_x3s: 0 off, 1 where gemm_f16's SPLIT build served, 2 every size)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine

eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
crops = synth.smooth_crops_u8(1024, 3)
for n in (1, 7, 30, 33, 64, 130, 256, 1024):
    x = crops[:n]
    eng.set_precision(0)
    exact = eng.embed_u8(x)
    eng.set_precision(2)
    out = {}
    for sw in (0, 1, 2):
        eng.debug_switch("conv_x3s", sw)
        out[sw] = eng.embed_u8(x)
    eng.debug_switch("conv_x3s", 1)
    sc = np.abs(exact).max()
    print("n=%4d  |x3s1 - off| %.2e  |x3s2 - off| %.2e   vs exact fp32: off %.2e  x3s1 %.2e  x3s2 %.2e   finite %s  fault %d"
          % (n, np.abs(out[1] - out[0]).max() / sc, np.abs(out[2] - out[0]).max() / sc, np.abs(out[0] - exact).max() / sc,
             np.abs(out[1] - exact).max() / sc, np.abs(out[2] - exact).max() / sc, np.isfinite(out[2]).all(), eng.fault_bits()))
d = parallel.DevArray.from_numpy(eng, crops)
emb = parallel.DevArray(eng, (1024, 512))
for n in (30, 256, 1024):
    for sw in (0, 1, 2):
        eng.debug_switch("conv_x3s", sw)
        for _ in range(3):
            eng.embed_u8_dev(d.ptr, n, emb.ptr)
        eng.timer_start()
        for _ in range(10):
            eng.embed_u8_dev(d.ptr, n, emb.ptr)
        print("n=%4d conv_x3s=%d: %.3f ms per pass" % (n, sw, eng.timer_stop() / 10))
eng.debug_switch("conv_x3s", 1)
