"""Four Swin-T passes of 256 images for rocprofv3 - which kernels carry a pass in a given mode:
    cd /tmp && rocprofv3 --kernel-trace -d out -o p -- python3 $REPO/tools/profile_swin_pass.py [precision]; python3 tools/rocprof_summary.py out/p_results.db"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import parallel, synth, weights
from reid_amd.engine import get_engine
eng = get_engine(0)
eng.load_swin(*weights.pack_swin(synth.swin_state_dict(0))[:2])
dx = parallel.DevArray.from_numpy(eng, synth.images_f32(256, 2))
emb = parallel.DevArray(eng, (256, 96))
eng.set_chunk(256)
eng.set_precision(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
for _ in range(4):
    eng.swin_embed_dev(dx.ptr, 256, 224, 224, emb.ptr)
eng.sync()
