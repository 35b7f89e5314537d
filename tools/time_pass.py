"""Time of one embed pass (device-resident uint8 crops, HIP events): python tools/time_pass.py [precision 0|1|2] [crops] - for A/B runs of
a debug switch inside ONE gpurun call (boxes differ by a few percent):
    for v in 0 1 0 1; do REID_DEBUG_SWITCHES=split_pair=$v python3 tools/time_pass.py; done
(REID_DEBUG_SWITCHES is read HERE and applied through libreid_hip_debug.so - Engine.debug_switches_from_env; the product library takes no
such switch from the environment.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = get_engine(0)
sw = eng.debug_switches_from_env()
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
crops = parallel.DevArray.from_numpy(eng, synth.crops_u8(n, 1))
emb = parallel.DevArray(eng, (n, 512))
eng.set_chunk(min(n, 1024))
eng.set_precision(prec)
for _ in range(3):
    eng.embed_u8_dev(crops.ptr, n, emb.ptr)
best = 1e9
for _ in range(3):
    eng.timer_start()
    for _ in range(5):
        eng.embed_u8_dev(crops.ptr, n, emb.ptr)
    best = min(best, eng.timer_stop() / 5)
print("precision %d, %d crops%s: %.3f ms per pass = %.1f k crops/s" % (prec, n, (" [" + sw + "]") if sw else "", best, n / best))
