"""ms per embed pass (mode 2, device-resident crops) for every pass size of a tracking stream: python tools/probes/small_sweep.py [lo] [hi] [step]
Weighted with the Poisson(30) frame sizes of bench.py's tracking workload: where the stream's kernel time goes."""
import math
import os
import sys

sys.path.insert(0, ".")
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 64
step = int(sys.argv[3]) if len(sys.argv) > 3 else 1
eng = get_engine(0)
sw = eng.debug_switches_from_env()
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
crops = parallel.DevArray.from_numpy(eng, synth.crops_u8(hi, 1))
emb = parallel.DevArray(eng, (hi, 512))
eng.set_chunk(1024)
eng.set_precision(2)
res = {}
for n in range(lo, hi + 1, step):
    for _ in range(3):
        eng.embed_u8_dev(crops.ptr, n, emb.ptr)
    best = 1e9
    for _ in range(3):
        eng.timer_start()
        for _ in range(8):
            eng.embed_u8_dev(crops.ptr, n, emb.ptr)
        best = min(best, eng.timer_stop() / 8)
    res[n] = best
tot = 0.0
wsum = 0.0
for n in range(lo, hi + 1, step):
    w = math.exp(-30.0 + n * math.log(30.0) - math.lgamma(n + 1.0))
    tot += w * res[n]
    wsum += w
    print("%3d crops %7.1f us  %5.2f us/crop  poisson %.4f" % (n, res[n] * 1e3, res[n] * 1e3 / n, w))
print("Poisson(30)-weighted mean over [%d, %d] (mass %.3f): %.1f us per frame%s" % (lo, hi, wsum, tot / wsum * 1e3, (" [" + sw + "]") if sw else ""))
