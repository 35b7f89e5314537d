"""A/B of an environment switch on the fp32-class forward: embeddings must be BIT-identical, time per pass of both.
python tools/epi_ab.py ENV_NAME [crops]"""
import os, subprocess, sys
import numpy as np
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine
n = %d
eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
crops = parallel.DevArray.from_numpy(eng, synth.crops_u8(n, 1))
emb = parallel.DevArray(eng, (n, 512))
eng.set_chunk(min(n, 1024)); eng.set_precision(2)
for _ in range(3): eng.embed_u8_dev(crops.ptr, n, emb.ptr)
best = 1e9
for _ in range(3):
    eng.timer_start()
    for _ in range(5): eng.embed_u8_dev(crops.ptr, n, emb.ptr)
    best = min(best, eng.timer_stop() / 5)
np.save(sys.argv[1], emb.numpy())
small = synth.crops_u8(77, 3)
eng.set_chunk(64)
np.save(sys.argv[1] + ".small.npy", eng.embed_u8(small))
print("%%.3f" %% best)
''' % (ROOT, n)
res = {}
for v in ("0", "1", "0", "1"):
    out = "/tmp/epi_ab_%s.npy" % v
    r = subprocess.run([sys.executable, "-c", child, out], capture_output=True, text=True, env=dict(os.environ, **{name: v}))
    print("%s=%s: %s ms per pass %s" % (name, v, r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else ""))
a, b = np.load("/tmp/epi_ab_0.npy"), np.load("/tmp/epi_ab_1.npy")
a2, b2 = np.load("/tmp/epi_ab_0.npy.small.npy"), np.load("/tmp/epi_ab_1.npy.small.npy")
print("bit-identical embeddings (%d crops, chunk 1024):" % n, bool(np.array_equal(a, b)), " (77 crops, chunk 64):", bool(np.array_equal(a2, b2)),
      " max abs diff", float(np.abs(a - b).max()), float(np.abs(a2 - b2).max()))
