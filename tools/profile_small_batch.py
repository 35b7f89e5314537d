"""One forward of a tracking-sized batch (30 crops) for rocprofv3: which kernels carry the frame latency.
    cd /tmp && rocprofv3 --kernel-trace --stats -d out -o p -- python3 $REPO/tools/profile_small_batch.py [n] [f16|f32|f16x3]
    python3 tools/timeline.py out/p_results.db <launches per forward>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd.engine import get_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
eng = get_engine(0)
sw = eng.debug_switches_from_env()       # A/B: REID_DEBUG_SWITCHES=split_x3_small=1
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0, gem_p=3.0))[:2])
eng.set_precision({"f32": 0, "f16": 1, "f16x3": 2}[prec])
crops = synth.smooth_crops_u8(n, 1)
for _ in range(5):
    eng.embed_u8(crops)
t0 = time.perf_counter()
for _ in range(50):
    eng.embed_u8(crops)
print("embed_u8(%d crops, %s%s) host round trip: %.0f us" % (n, prec, (" [" + sw + "]") if sw else "", (time.perf_counter() - t0) / 50 * 1e6))
