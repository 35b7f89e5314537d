"""Where a pipelined tracking frame spends its time on the host: pack + submit | wait for the costs | update.
A long wait means the device is the bottleneck, a short one the host.  python tools/track_pipeline_split.py [f16|f32]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import get_engine
from reid_amd.nn_matching import NearestNeighborDistanceMetric

eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0, gem_p=3.0))[:2])
eng.set_precision(1 if (sys.argv[1:] or ["f16"])[0] == "f16" else 0)
rng = np.random.default_rng(3)
frames = 400
counts = np.clip(rng.poisson(30, frames), 1, 80)
pool = synth.ragged_crops_u8(256, seed=3)
print("mean crop bytes %.0f, crops/frame %.1f" % (np.mean([c.size for c in pool]), counts.mean()))
metric = NearestNeighborDistanceMetric("cosine", 0.15, 100)
tracks = list(range(40))
metric.partial_fit(rng.normal(size=(4000, 512)).astype(np.float32), np.repeat(tracks, 100), tracks)
boxes = rng.uniform(0, 500, (80, 4))
boxes[:, 2:] = rng.uniform(20, 120, (80, 2))
crops_of = lambda f: [pool[(f * 7 + i) % 256] for i in range(int(counts[f]))]
T = np.zeros(4)
eng.frame_submit(0, crops_of(0))
t_all = time.perf_counter()
for f in range(frames):
    slot = f & 1
    t0 = time.perf_counter()
    n = int(counts[f])
    metric.frame_distance_begin(slot, tracks, 0.15, boxes[:40], boxes[:n])
    t1 = time.perf_counter()
    if f + 1 < frames:
        eng.frame_submit(slot ^ 1, crops_of(f + 1))
    t2 = time.perf_counter()
    metric.frame_distance_end(slot)
    t3 = time.perf_counter()
    k = min(n, 40)
    metric.frame_partial_fit(slot, np.arange(k, dtype=np.int32), tracks[:k], tracks)
    t4 = time.perf_counter()
    if f >= 50:
        T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]
eng.sync()
print("per frame us: enqueue costs %.0f | pack + submit next %.0f | wait for costs %.0f | update %.0f   (%.0f frames/s)"
      % (*(T / (frames - 50) * 1e6), frames / (time.perf_counter() - t_all)))
