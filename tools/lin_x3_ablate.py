"""Timing ablations of lin_x3_kernel (debug switch x3_ablate: 1 no weight DMA in the loop, 2 no A DMA, 4 no fragment reads, 32 no epilogue; results are
then wrong) on the Swin stage-3 / stage-4 linear shapes at n images: python tools/lin_x3_ablate.py [n=512]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng = get_engine(0)
eng.load_swin(*weights.pack_swin(synth.swin_state_dict(0))[:2])
eng.set_precision(2)
rng = np.random.default_rng(0)
big = parallel.DevArray.from_numpy(eng, rng.normal(size=(n, 3, 224, 224)).astype(np.float32))
out = parallel.DevArray(eng, (n, 96))
base = None
def run_pass():
    try:
        eng.swin_embed_dev(big.ptr, n, 224, 224, out.ptr)
    except Exception:        # the fault word raised by the wrong values of an ablated run: clear it, the launches were all made
        eng.clear_fault()


def sync():
    try:
        eng.sync()
    except Exception:
        eng.clear_fault()


for abl in (0, 1, 2, 3, 4, 7, 32, 39, 0):
    eng.debug_switch("x3_ablate", abl)
    for _ in range(2):
        run_pass()
    sync()
    t0 = time.perf_counter()
    for _ in range(3):
        run_pass()
    sync()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    base = base or ms
    print("x3_ablate=%2d: %d images %.2f ms (%+.2f ms)" % (abl, n, ms, ms - base), flush=True)
eng.debug_switch("x3_ablate", 0)
eng.clear_fault()
