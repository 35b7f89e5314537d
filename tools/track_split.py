"""Host-time split of one tracking frame driven call by call (embed_ragged / bank cost / DIoU / bank update, each blocking).
python tools/track_split.py"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from reid_amd import synth, weights
from reid_amd.engine import get_engine
from reid_amd.iou_matching import iou_cost
from reid_amd.nn_matching import NearestNeighborDistanceMetric
eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0, gem_p=3.0))[:2])
eng.set_precision(1)
rng = np.random.default_rng(3)
pool = synth.ragged_crops_u8(256, seed=3)
metric = NearestNeighborDistanceMetric("cosine", 0.15, 100)
tracks = list(range(40))
metric.partial_fit(rng.normal(size=(4000, 512)).astype(np.float32), np.repeat(tracks, 100), tracks)
boxes = rng.uniform(0, 500, (80, 4)); boxes[:, 2:] = rng.uniform(20, 120, (80, 2))
T = np.zeros(4)
for f in range(300):
    crops = [pool[(f * 7 + i) % 256] for i in range(30)]
    t0 = time.perf_counter(); feats = eng.embed_ragged_u8(crops)
    t1 = time.perf_counter(); cost = metric.distance(feats, tracks, max_distance=0.15)
    t2 = time.perf_counter(); ic = iou_cost(boxes[:40], boxes[:30])
    t3 = time.perf_counter(); metric.partial_fit(feats[:30], tracks[:30], tracks)
    t4 = time.perf_counter()
    if f >= 20: T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]
print("per frame us: embed_ragged %.0f, bank cost %.0f, diou %.0f, bank update %.0f" % tuple(T / 280 * 1e6))
