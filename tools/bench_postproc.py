"""Evaluation post-processing at the reference's size on one MI355X: camera de-biasing of 19 281 x 1 263 descriptors over 6
cameras (inference_utils.py:5-15; Market-1501 gallery + query) and the flip-TTA descriptor of 1 024 images, against the
CPU oracle.  python tools/bench_postproc.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import postproc
from reid_amd import synth, weights
from reid_amd.engine import get_engine

eng = get_engine(0)
_, _, _, x, _, cams = synth.clustered_embeddings(1, 19281, d=1263, n_ids=751, n_cams=6, seed=6, sigma=0.9)
eng.cam_debias(x[:2000], cams[:2000])
t0 = time.perf_counter()
got = eng.cam_debias(x, cams)
gpu_ms = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
want = postproc.diminish_camera_bias(x, cams)
cpu_ms = (time.perf_counter() - t0) * 1e3
res = {"workload": "diminish_camera_bias 19281 x 1263, 6 cameras (incl. H2D/D2H of 97 MB each way)", "gpu_ms": round(gpu_ms, 1),
       "cpu_oracle_ms": round(cpu_ms, 1), "cores": len(os.sched_getaffinity(0)), "max_abs_diff": float(np.abs(got - want).max())}
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
eng.set_precision(1)
eng.set_chunk(1024)
imgs = np.random.default_rng(0).normal(0, 1, (1024, 3, 256, 128)).astype(np.float32)
eng.descriptor_f32_nchw(imgs[:64])
t0 = time.perf_counter()
d = eng.descriptor_f32_nchw(imgs, flip_tta=True)
res["tta_descriptor_1024_images_ms_incl_copies"] = round((time.perf_counter() - t0) * 1e3, 1)
print(json.dumps(res))
