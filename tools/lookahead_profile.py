"""Where a look-ahead group's time goes (tracking.LookaheadCameraStream): python tools/lookahead_profile.py [F] [match_stream 0|1] [idle contexts created first]"""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from reid_amd import synth, weights
from reid_amd.tracking import LookaheadCameraStream

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
MS = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
sd = synth.seres18_state_dict(0, gem_p=3.0)
blob, manifest = weights.pack_seres18(sd)[:2]
rng = np.random.default_rng(3)
frames = 400
counts = np.clip(rng.poisson(30, frames), 1, 80)
pool = synth.ragged_crops_u8(256, seed=3)
tracks = list(range(40))
boxes = rng.uniform(0, 500, (80, 4))
boxes[:, 2:] = rng.uniform(20, 120, (80, 2))
from reid_amd.engine import Engine
idle = [Engine(0) for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 0)]     # (their streams share the hardware queues)
la = LookaheadCameraStream(blob, manifest, F, 2, match_stream=MS)
la.metric.partial_fit(rng.normal(size=(40 * 100, 512)).astype(np.float32), np.repeat(tracks, 100), tracks)
crops_of = lambda f: [pool[(f * 7 + i) % 256] for i in range(int(counts[f]))]
t_step = []


def drive(first, last):
    grp = lambda g0: [crops_of(f) for f in range(g0, min(g0 + F, last))]
    la.submit_group(grp(first))
    for g0 in range(first, last, F):
        for j, f in enumerate(range(g0, min(g0 + F, last))):
            n = int(counts[f])
            t0 = time.perf_counter()
            la.step(j, tracks, boxes[:40], boxes[:n], grp(g0 + F) if (j == la.handover and g0 + F < last) else None)
            t_step.append((j, time.perf_counter() - t0))
            k = min(n, 40)
            la.commit(j, np.arange(k), tracks[:k], tracks)
    la.eng.sync()


drive(0, 40)
t_step.clear()
t0 = time.perf_counter()
drive(0, frames)
el = time.perf_counter() - t0
print("F=%d match_stream=%d: %.3f ms per frame = %.0f frames/s" % (F, MS, el / frames * 1e3, frames / el))
for j in range(F):
    v = [t for jj, t in t_step if jj == j]
    print("  step(j=%d): mean %.3f ms" % (j, np.mean(v) * 1e3))
pr = cProfile.Profile()
pr.enable()
drive(0, 100)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
