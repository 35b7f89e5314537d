"""Several camera streams on one GPU: one engine context (own HIP stream, workspaces, feature bank) and one host thread per
camera, each running the frame pipeline.  A tracking frame leaves most CUs idle in most launches, so independent streams
overlap on the device.  python tools/track_cameras.py [cameras] [f16|f32]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import Engine
from reid_amd.nn_matching import NearestNeighborDistanceMetric

cams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
frames = 400
blob, manifest = weights.pack_seres18(synth.seres18_state_dict(0, gem_p=3.0))[:2]
pool = synth.ragged_crops_u8(256, seed=3)


class Camera:
    def __init__(self, idx):
        self.eng = Engine(0)
        self.eng.load_seres18(blob, manifest)
        self.eng.set_precision(1 if prec == "f16" else 0)
        rng = np.random.default_rng(3 + idx)
        self.counts = np.clip(rng.poisson(30, frames), 1, 80)
        self.metric = NearestNeighborDistanceMetric("cosine", 0.15, 100, engine=self.eng)
        self.tracks = list(range(40))
        self.metric.partial_fit(rng.normal(size=(4000, 512)).astype(np.float32), np.repeat(self.tracks, 100), self.tracks)
        self.boxes = rng.uniform(0, 500, (80, 4))
        self.boxes[:, 2:] = rng.uniform(20, 120, (80, 2))
        self.idx = idx

    def crops_of(self, f):
        return [pool[(f * 7 + i + 31 * self.idx) % 256] for i in range(int(self.counts[f]))]

    def run(self, first, last):
        eng, metric = self.eng, self.metric
        eng.frame_submit(first & 1, self.crops_of(first))
        for f in range(first, last):
            slot, n = f & 1, int(self.counts[f])
            metric.frame_distance_begin(slot, self.tracks, 0.15, self.boxes[:40], self.boxes[:n])
            if f + 1 < last:
                eng.frame_submit(slot ^ 1, self.crops_of(f + 1))
            metric.frame_distance_end(slot)
            k = min(n, 40)
            metric.frame_partial_fit(slot, np.arange(k, dtype=np.int32), self.tracks[:k], self.tracks)
        eng.sync()


cameras = [Camera(i) for i in range(cams)]
for c in cameras:
    c.run(0, 60)                       # warm-up: workspaces, graphs
threads = [threading.Thread(target=c.run, args=(0, frames)) for c in cameras]
t0 = time.perf_counter()
for t in threads:
    t.start()
for t in threads:
    t.join()
el = time.perf_counter() - t0
print("%d camera stream(s), %s: %.0f frames/s in total (%.0f per camera, %.2f ms per frame and camera)"
      % (cams, prec, cams * frames / el, frames / el, el / frames * 1e3))
