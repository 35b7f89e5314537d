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
crops = synth.smooth_crops_u8(9, 3)
eng.set_precision(2)
eng.debug_switch("split_x3_min_blocks", 1)
eng.debug_switch("f16_split_k", 0)
eng.debug_switch("x3_unroll", 0); a = eng.embed_u8(crops)
eng.debug_switch("x3_unroll", 1); b = eng.embed_u8(crops)
print("9 crops: unrolled == looped form: %s  (max rel %.2e)" % (np.array_equal(a, b), float(np.abs(a - b).max() / np.abs(a).max())))
big = synth.crops_u8(1024, 1)
eng.debug_switch("split_x3_min_blocks", 512); eng.debug_switch("f16_split_k", 1)
eng.debug_switch("x3_unroll", 0); a = eng.embed_u8(big)
eng.debug_switch("x3_unroll", 1); b = eng.embed_u8(big)
print("1024 crops: unrolled == looped form: %s  (max rel %.2e)" % (np.array_equal(a, b), float(np.abs(a - b).max() / np.abs(a).max())))
PY
for sw in "x3_unroll=0" "x3_unroll=1" "x3_unroll=0" "x3_unroll=1"; do
  REID_DEBUG_SWITCHES=$sw timeout -k 5 120 python tools/time_pass.py 2 1024 2>&1 | tail -1
done
} > gpurun_out/r5/x3u.txt 2>&1
cat gpurun_out/r5/x3u.txt
