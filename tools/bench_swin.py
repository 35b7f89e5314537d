"""BASELINE config 3: Swin-T (v1) embedding throughput at 224x224 on one MI355X (fp32 MFMA path), with the CPU oracle
timed beside it.  python tools/bench_swin.py [n_images] [chunk]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import _ffi, synth, weights
from reid_amd.engine import get_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
eng = get_engine(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
eng.set_stream(stream.cuda_stream)
eng.set_chunk(chunk)
sd = synth.swin_state_dict(0)
eng.load_swin(*weights.pack_swin(sd)[:2])
eng.set_precision(1 if prec == "f16" else 0)
x = torch.from_numpy(synth.images_f32(64, 1)).cuda().repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
emb = torch.empty((n, 96), dtype=torch.float32, device="cuda")
for _ in range(2):
    eng.swin_embed_dev(x.data_ptr(), n, 224, 224, emb.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 3
for _ in range(steps):
    eng.swin_embed_dev(x.data_ptr(), n, 224, 224, emb.data_ptr())
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / steps
eng.profile_reset()
eng.profile(True)
eng.swin_embed_dev(x.data_ptr(), n, 224, 224, emb.data_ptr())
torch.cuda.synchronize()
g, e = eng.profile_get(_ffi.K_CONV_GEMM), eng.profile_get(_ffi.K_ELEMENTWISE)
eng.profile(False)
out = {"workload": "BASELINE configs[2]: Swin-T v1, %d images 224x224, %s" % (n, "fp16-storage GEMMs / fp32 residual stream" if prec == "f16" else "fp32 MFMA GEMMs"), "crops_per_s": round(n / el, 1),
       "ms": round(el * 1e3, 2), "gemm_tflops": round(g["flops"] / g["ms"] / 1e9, 2), "gemm_ms": round(g["ms"], 2),
       "gemm_launches": g["launches"], "other_ms": round(e["ms"], 2), "chunk": chunk,
       "precision": prec, "whole_net_tflops": round(11.54e9 * n / el / 1e12, 1)}
if "--cpu" in sys.argv:
    from oracle import swin
    torch.set_num_threads(16)
    xs = x[:16].cpu().numpy()
    swin.embed(sd, xs)
    t0 = time.perf_counter()
    m = 0
    while time.perf_counter() - t0 < 10:
        swin.embed(sd, xs)
        m += 16
    out["cpu_oracle_crops_per_s"] = round(m / (time.perf_counter() - t0), 2)
print(json.dumps(out))
