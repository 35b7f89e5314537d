"""Throughput of the fp16 embed path as a function of the batch size (device-resident crops): shows where the small-batch
heuristics (tile-parallel vs per-image kernels, 64- vs 128-wide halo tiles) hand over.  python tools/bench_batch_sweep.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import get_engine

eng = get_engine(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
eng.set_stream(stream.cuda_stream)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
eng.set_precision(1)
eng.set_chunk(1024)
base = torch.from_numpy(synth.crops_u8(256, seed=1)).cuda()
out = {}
for n in (8, 16, 30, 64, 96, 127, 128, 192, 256, 512, 1024, 2048):
    crops = base.repeat((n + 255) // 256, 1, 1, 1)[:n].contiguous()
    emb = torch.empty((n, 512), dtype=torch.float32, device="cuda")
    for _ in range(3):
        eng.embed_u8_dev(crops.data_ptr(), n, emb.data_ptr())
    torch.cuda.synchronize()
    reps = max(3, min(200, 20000 // n))
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.embed_u8_dev(crops.data_ptr(), n, emb.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out[n] = {"ms": round(dt * 1e3, 3), "crops_per_s": round(n / dt, 0)}
print(json.dumps({"workload": "fp16 embed, device-resident crops, by batch size", "by_n": out}))
