"""Where a batched frame time of K cameras goes: host (packing, Python) against device (python tools/multicam_profile.py [K])."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from reid_amd import synth, weights
from reid_amd.tracking import MultiCameraStream

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sd = synth.seres18_state_dict(0, gem_p=3.0)
blob, manifest = weights.pack_seres18(sd)[:2]
rng = np.random.default_rng(3)
frames = 300
counts = np.clip(rng.poisson(30, frames), 1, 80)
pool = synth.ragged_crops_u8(256, seed=3)
tracks = list(range(40))
boxes = rng.uniform(0, 500, (80, 4))
boxes[:, 2:] = rng.uniform(20, 120, (80, 2))
mc = MultiCameraStream(blob, manifest, K, 2)
for met in mc.metrics:
    met.partial_fit(rng.normal(size=(40 * 100, 512)).astype(np.float32), np.repeat(tracks, 100), tracks)


def crk(f):
    return [[pool[(f * 7 + i + 31 * c) % 256] for i in range(int(counts[f]))] for c in range(K)]


def drive(first, last):
    mc.submit(crk(first))
    for f in range(first, last):
        n = int(counts[f])
        mc.step([tracks] * K, [boxes[:40]] * K, [boxes[:n]] * K, crk(f + 1) if f + 1 < last else None)
        k = min(n, 40)
        mc.commit([np.arange(k)] * K, [tracks[:k]] * K, [tracks] * K)
    mc.eng.sync()


drive(0, 40)
t0 = time.perf_counter()
drive(0, frames)
el = time.perf_counter() - t0
print("K=%d: %.3f ms per frame time, %.1f frames/s total" % (K, el / frames * 1e3, K * frames / el))
# device time of one pass of the same size, back to back (no host work in between)
crops = [c for cam in crk(5) for c in cam]
eng = mc.eng
eng.embed_ragged_u8(crops)
t0 = time.perf_counter()
for _ in range(50):
    eng.embed_ragged_u8(crops)
print("synchronous embed of %d ragged crops: %.3f ms per call" % (len(crops), (time.perf_counter() - t0) / 50 * 1e3))
pr = cProfile.Profile()
pr.enable()
drive(0, 100)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
