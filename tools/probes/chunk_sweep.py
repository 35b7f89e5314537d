import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine
eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
n = 4096
crops = parallel.DevArray.from_numpy(eng, synth.crops_u8(n, 1))
emb = parallel.DevArray(eng, (n, 512))
eng.set_precision(2)
for chunk in (1024, 2048, 4096, 1024, 2048):
    eng.set_chunk(chunk)
    for _ in range(2):
        eng.embed_u8_dev(crops.ptr, n, emb.ptr)
    best = 1e9
    for _ in range(3):
        eng.timer_start()
        for _ in range(3):
            eng.embed_u8_dev(crops.ptr, n, emb.ptr)
        best = min(best, eng.timer_stop() / 3)
    print("chunk %d: %.3f ms per %d crops = %.1f k crops/s" % (chunk, best, n, n / best), flush=True)
