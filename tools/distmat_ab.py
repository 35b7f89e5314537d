"""Times the Market-size distance matrix (3368 x 15913 x d, euclidean) on device buffers, per value of one debug switch:
    python tools/distmat_ab.py [d=512] [switch=x3_ablate] [values=0,1,2,3]
(the switch is only a channel for kernel experiments; the product library reads no environment)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi
from reid_amd.engine import get_engine

d = int(sys.argv[1]) if len(sys.argv) > 1 else 512
name = sys.argv[2] if len(sys.argv) > 2 else "x3_ablate"
values = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2,3").split(",")]
m, n = 3368, 15913
eng = get_engine(0)
g = torch.Generator().manual_seed(0)
x = torch.randn(m, d, generator=g).cuda()
y = torch.randn(n, d, generator=g).cuda()
out = torch.empty(m, n, device="cuda")
ref = None
for rep in range(2):
    for v in values:
        eng.debug_switch(name, v)
        for _ in range(3):
            eng.distmat_dev(x.data_ptr(), m, y.data_ptr(), n, d, _ffi.METRIC_L2, out.data_ptr())
        eng.sync()
        t0 = time.perf_counter()
        iters = 20
        for _ in range(iters):
            eng.distmat_dev(x.data_ptr(), m, y.data_ptr(), n, d, _ffi.METRIC_L2, out.data_ptr())
        eng.sync()
        us = (time.perf_counter() - t0) / iters * 1e6
        got = out.cpu().numpy()
        if ref is None:
            ref = got
        print("%s=%d: %.1f us  = %.1f TF (%.3f of 157.3)   max |diff| vs first %.3g" % (name, v, us, 2.0 * m * n * d / us / 1e6, 2.0 * m * n * d / us / 1e6 / 157.3, np.abs(got - ref).max()), flush=True)
eng.debug_switch(name, 0)
