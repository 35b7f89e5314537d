"""Five passes of 1024 crops in precision 2 (fp32-class) for rocprofv3 - which kernels carry the pass:
    cd /tmp && rocprofv3 --kernel-trace --stats -d out -o p -- python3 $REPO/tools/profile_split_pass.py [precision]; python3 tools/rocprof_summary.py out/p_results.db"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine

eng = get_engine(0)
eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0))[:2])
crops = parallel.DevArray.from_numpy(eng, synth.crops_u8(1024, 1))
emb = parallel.DevArray(eng, (1024, 512))
eng.set_chunk(1024)
eng.set_precision(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
for _ in range(5):
    eng.embed_u8_dev(crops.ptr, 1024, emb.ptr)
eng.sync()
