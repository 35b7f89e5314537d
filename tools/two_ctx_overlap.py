"""Do two contexts embedding half the batch each (two HIP streams, two workspaces, two host threads) beat one context embedding all of it?
An elementwise pass of one (HBM-bound, no MFMA) could run beside a convolution of the other.  python tools/two_ctx_overlap.py [crops]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import parallel, synth, weights
from reid_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blob, manifest = weights.pack_seres18(synth.seres18_state_dict(0, gem_p=3.0))[:2]
engs = [Engine(0), Engine(0)]
bufs = []
for e in engs:
    e.load_seres18(blob, manifest)
    e.set_precision(2)
    e.set_chunk(1024)
    bufs.append((parallel.DevArray.from_numpy(e, synth.crops_u8(n, 1)), parallel.DevArray(e, (n, 512))))

def run(i, m, reps):
    e, (c, o) = engs[i], bufs[i]
    for _ in range(reps):
        e.embed_u8_dev(c.ptr, m, o.ptr)
    e.sync()

run(0, n, 2); run(1, n // 2, 2)
t0 = time.perf_counter(); run(0, n, 5); one = (time.perf_counter() - t0) / 5
th = [threading.Thread(target=run, args=(i, n // 2, 5)) for i in range(2)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
two = (time.perf_counter() - t0) / 5
print("%d crops: one context %.2f ms = %.1f k crops/s; two contexts x %d crops concurrently %.2f ms = %.1f k crops/s" % (n, one * 1e3, n / one / 1e3, n // 2, two * 1e3, n / two / 1e3))
