"""Where does an image's Swin result start to depend on its POSITION in the pass?  Copies of three images are spread over a pass;
the debug switch swin_stop freezes the forward after a phase of a block (block * 10 + phase: 0 = attention branch done, 2 = LayerNorm 2 written,
3 = fc1 + GELU written, 5 = whole block) and the scratch buffers are compared between copies, element by element.  Prints the first
phase whose buffer differs, with the differing elements' (token, column), their tile row (token index mod 256) and both values.
    python tools/swin_position_probe.py [precision] [n_images] [blocks]"""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reid_amd import synth, weights
from reid_amd._ffi import check
from reid_amd.engine import Engine

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 2
blob, manifest = weights.pack_swin(synth.swin_state_dict(0))[:2]
base = synth.images_f32(3, 2)
ids = np.asarray([(i * 7 + i // 3) % 3 for i in range(n)])
x = base[ids]
first = [int(np.flatnonzero(ids == c)[0]) for c in range(3)]
TOK = 3136


STAGE_OF_BLOCK = [0, 0, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3]


def run(stop, stage):
    eng = Engine(0)
    eng.debug_switch("swin_stop", stop)
    eng.load_swin(blob, manifest)
    eng.set_precision(prec)
    eng.set_chunk(256)
    eng.debug_keep(True)
    eng.swin_embed_f32_nchw(x)
    tok, ch = TOK >> (2 * stage), 96 << stage
    out = {}
    for st, per in ((1 + stage, tok * ch), (6, tok * ch), (7, tok * ch * 4)):
        full = TOK * 96 if st == 6 else TOK * 384 if st == 7 else tok * ch     # floats per image the entry point reports
        buf = np.empty(n * full, np.float32)
        cnt = C.c_size_t()
        check(eng.lib.reid_debug_swin_stage(eng.h, st, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(cnt)))
        assert cnt.value == buf.size, (st, cnt.value, buf.size)
        out[st] = buf
    eng.close()
    return out


def as_rows(buf, tok, cols, f16):
    """scratch buffer of `n` images -> [n][tok][cols] in the element type the forward wrote (the valid prefix of the buffer)"""
    raw = buf.reshape(-1).view(np.uint8)
    if f16:
        return raw[: n * tok * cols * 2].view(np.float16).reshape(n, tok, cols)
    return raw[: n * tok * cols * 4].view(np.float32).reshape(n, tok, cols)


def compare(name, t):
    bad = 0
    tok = t.shape[1]
    for i in range(n):
        ref = t[first[ids[i]]]
        d = np.argwhere(t[i] != ref)
        bad += len(d)
        for (tk, col) in d[:3]:
            g_a, g_b = i * tok + tk, first[ids[i]] * tok + tk
            print("   %s: image %d (copy of %d) token %d col %d: %r vs %r   tile rows %d / %d"
                  % (name, i, first[ids[i]], tk, col, t[i][tk, col], ref[tk, col], g_a % 256, g_b % 256))
    print("%-52s differing elements: %d of %d" % (name, bad, t.size), flush=True)
    return bad


b0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
for b in range(b0, b0 + blocks):
    st = STAGE_OF_BLOCK[b]
    tok, ch = TOK >> (2 * st), 96 << st
    for phase, what in ((0, "x after attention branch"), (2, "LayerNorm-2 output"), (3, "fc1 + GELU output"), (5, "x after the block")):
        o = run(b * 10 + phase, st)
        if phase in (0, 5):
            compare("block %d: %s (fp32 stream)" % (b, what), as_rows(o[1 + st], tok, ch, False))
        elif phase == 2:
            compare("block %d: %s" % (b, what), as_rows(o[6], tok, ch * (2 if prec == 2 else 1), prec != 0))
        else:
            compare("block %d: %s" % (b, what), as_rows(o[7], tok, 4 * ch * (2 if prec == 2 else 1), prec != 0))
