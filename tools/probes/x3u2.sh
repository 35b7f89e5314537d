mkdir -p gpurun_out/r5
{
python - <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from reid_amd import synth, weights
from reid_amd.engine import get_engine
eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
eng.set_precision(2)
big = synth.crops_u8(1024, 1)
eng.debug_switch("x3_unroll", 1); a = eng.embed_u8(big)
for v in (3, 5):
    eng.debug_switch("x3_unroll", v); b = eng.embed_u8(big)
    print("1024 crops: x3_unroll=%d == 1: %s  (max rel %.2e)" % (v, np.array_equal(a, b), float(np.abs(a - b).max() / np.abs(a).max())))
PY
for sw in "x3_unroll=3" "x3_unroll=5" "x3_unroll=3" "x3_unroll=5"; do
  REID_DEBUG_SWITCHES=$sw timeout -k 5 120 python tools/time_pass.py 2 1024 2>&1 | tail -1
done
} > gpurun_out/r5/x3u2.txt 2>&1
cat gpurun_out/r5/x3u2.txt
