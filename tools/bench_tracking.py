"""BASELINE config 4 stand-in (MOT16-02 is not in the container): 600 synthetic frames, detections/frame ~ Poisson(30)
clipped to [1,80], crop heights log-uniform in [40,400], w = h*U(0.3,0.5).  Per frame, through the
plugin surface: Extractor(crops) -> cosine cost against a 100-feature bank per track -> DIoU cost.
python tools/bench_tracking.py [frames] [f16|f32]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi, synth
from reid_amd.extractor import Extractor
from reid_amd.iou_matching import iou_cost
from reid_amd.nn_matching import NearestNeighborDistanceMetric

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
rng = np.random.default_rng(3)
ext = Extractor(synth.seres18_state_dict(0, gem_p=3.0))
eng = ext.net
eng.set_precision(1 if prec == "f16" else 0)
counts = np.clip(rng.poisson(30, frames), 1, 80)
pool = synth.ragged_crops_u8(256, seed=3)
metric = NearestNeighborDistanceMetric("cosine", 0.15, 100)         # MAX_DIST / NN_BUDGET, deep_sort.yaml:3,9
tracks = list(range(40))
metric.partial_fit(rng.normal(size=(40 * 100, 512)).astype(np.float32), np.repeat(tracks, 100), tracks)   # full banks
boxes = rng.uniform(0, 500, (80, 4))
boxes[:, 2:] = rng.uniform(20, 120, (80, 2))
for _ in range(3):
    ext(pool[:30])
lat = []
t0 = time.perf_counter()
ncrops = 0
for f in range(frames):
    n = int(counts[f])
    crops = [pool[(f * 7 + i) % 256] for i in range(n)]
    t1 = time.perf_counter()
    feats = ext(crops)                                               # feature_extractor.py:48-53
    cost = metric.distance(feats, tracks, max_distance=0.15)         # nn_matching.distance + min_cost_matching gate, one launch
    icost = iou_cost(boxes[:40], boxes[:n])                          # iou_matching.py:5-47 for every (track, detection)
    k = min(n, 40)
    metric.partial_fit(feats[:k], tracks[:k], tracks)                # matched tracks take this frame's feature into their ring
    lat.append(time.perf_counter() - t1)
    ncrops += n
el = time.perf_counter() - t0
lat = np.asarray(lat) * 1e3
print(json.dumps({"workload": "config 4 stand-in: %d frames, %d crops, ragged sizes, extractor + device feature-bank cost (40 tracks x 100) + DIoU + bank update" % (frames, ncrops),
                  "precision": prec, "frames_per_s": round(frames / el, 1), "crops_per_s": round(ncrops / el, 1),
                  "ms_per_frame_median": round(float(np.median(lat)), 3), "ms_per_frame_p95": round(float(np.percentile(lat, 95)), 3)}))
