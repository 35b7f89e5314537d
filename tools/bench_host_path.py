"""PCIe-inclusive rate of the host-buffer boundary: reid_embed_u8 on 4096 uint8 crops in host memory (H2D of the crops,
fp16 forward, D2H of the embeddings, one synchronous call) - the number DESIGN.md section 5 quotes beside `value`.
python tools/bench_host_path.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import get_engine

eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
eng.set_precision(1)
eng.set_chunk(1024)
crops = synth.crops_u8(4096, seed=1)
eng.embed_u8(crops[:1024])
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    eng.embed_u8(crops)
    best = min(best, time.perf_counter() - t0)
print(json.dumps({"workload": "reid_embed_u8, 4096 host crops (pageable numpy memory), fp16 mode, chunk 1024",
                  "ms": round(best * 1e3, 2), "crops_per_s_pcie_inclusive": round(4096 / best, 1),
                  "h2d_mb": round(crops.nbytes / 1e6, 1)}))
