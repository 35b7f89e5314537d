"""conv3x3_x3.hip (two blocks per CU) against conv3x3_f16.hip's SPLIT build and exact fp32: python tools/x3_check.py [crops]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
small = synth.smooth_crops_u8(10, 3)
eng.set_precision(0)
e0 = eng.embed_u8(small)
eng.set_precision(2)
eng.debug_switch("split_x3", 0)
e_old = eng.embed_u8(small)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
eng.debug_switch("split_x3_min_blocks", 1)
for form in (1, 2, 3):
    eng.debug_switch("split_x3", form)
    e_new = eng.embed_u8(small)
    print("10 crops, x3 form %d: old vs fp32 %.2e, new vs fp32 %.2e, new vs old %.2e" % (form, rel(e_old, e0), rel(e_new, e0), rel(e_new, e_old)))
eng.debug_switch("split_x3_min_blocks", 512)
crops = parallel.DevArray.from_numpy(eng, synth.crops_u8(n, 1))
emb = parallel.DevArray(eng, (n, 512))
eng.set_chunk(min(n, 1024))
res = {}
for rep in range(2):
    for sw in (0, 2, 3):
        eng.debug_switch("split_x3", sw)
        for _ in range(2):
            eng.embed_u8_dev(crops.ptr, n, emb.ptr)
        eng.timer_start()
        for _ in range(5):
            eng.embed_u8_dev(crops.ptr, n, emb.ptr)
        ms = eng.timer_stop() / 5
        res[sw] = emb.numpy()
        print("split_x3=%d: %.3f ms per %d crops = %.1f k crops/s" % (sw, ms, n, n / ms))
print("%d crops: form 2 vs old %.2e, form 3 vs old %.2e" % (n, rel(res[2], res[0]), rel(res[3], res[0])))
