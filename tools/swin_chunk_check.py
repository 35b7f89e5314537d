"""Swin embeddings of the same images under different pass sizes (REID_SWIN_CHUNK_MAX) must be bit-identical (images are
independent in eval mode; every kernel is position-invariant).  python tools/swin_chunk_check.py [images] [precision]"""
import json, os, subprocess, sys
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 2
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, numpy as np; sys.path.insert(0, %r); from reid_amd import synth, weights; from reid_amd.engine import get_engine;"
        "eng = get_engine(0); eng.load_swin(*weights.pack_swin(synth.swin_state_dict(0))[:2]); eng.set_chunk(4096); eng.set_precision(%d);"
        "x = synth.images_f32(64, 9); x = np.concatenate([x] * (%d // 64)); np.save(sys.argv[1], eng.swin_embed_f32_nchw(x))" % (root, prec, n))
out = {}
for cap in ("256", "512", "1024"):      # REID_SWIN_CHUNK_MAX lowers the library's cap of 1024
    path = "/tmp/swin_chunk_%s.npy" % cap
    subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, REID_SWIN_CHUNK_MAX=cap), check=True)
    out[cap] = np.load(path)
ref = out["256"]
print("finite", np.isfinite(ref).all(), "copies of the 64 images equal", all(np.array_equal(ref[:64], ref[i:i + 64]) for i in range(0, n, 64)))
for cap in ("512", "1024"):
    print("pass size %s vs 256: equal %s, max |diff| %.3e" % (cap, np.array_equal(out[cap], ref), np.abs(out[cap] - ref).max()))
