"""Seeded synthetic weights and crops (no checkpoint or dataset is reachable offline).

The reference's weights live on Google Drive (REID_EVAL.md:1) and its datasets
are not in the container, so benchmarks and parity tests use random-init
weights of the exact architecture.  Everything here is driven by
``numpy.random.default_rng(seed)`` in a fixed key order, so the same seed gives
the same ``state_dict`` on every machine (this container, the GPU box).

``seres18_state_dict`` emits exactly the keys/shapes of
``SERse18_IBN().state_dict()`` (reid/backbones/SERes18_IBN.py:186-248; key list
in SURVEY.md Appendix A).  ``oracle/gen_golden.py`` loads it into the
reference's own class with ``strict=True`` which pins the key layout.
"""
from collections import OrderedDict

import numpy as np

# (name, channels, has_ibn, has_downsample, in_channels)
SERES18_BLOCKS = [
    ("basicBlock11", 64, True, False, 64),
    ("basicBlock12", 64, True, False, 64),
    ("basicBlock21", 128, True, True, 64),
    ("basicBlock22", 128, True, False, 128),
    ("basicBlock31", 256, True, True, 128),
    ("basicBlock32", 256, True, False, 256),
    ("basicBlock41", 512, False, True, 256),
    ("basicBlock42", 512, False, False, 512),
]


def se_mid(c):
    """SE bottleneck width, reid/backbones/SERes18_IBN.py:17."""
    return max(8, c // 16)


def _bn(rng, sd, prefix, c):
    sd[prefix + ".weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    sd[prefix + ".bias"] = rng.normal(0.0, 0.1, c).astype(np.float32)
    sd[prefix + ".running_mean"] = rng.normal(0.0, 0.1, c).astype(np.float32)
    sd[prefix + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    sd[prefix + ".num_batches_tracked"] = np.asarray(0, dtype=np.int64)


def _conv(rng, cout, cin, k, gain=2.0):
    std = np.sqrt(gain / (cin * k * k))
    return rng.normal(0.0, std, (cout, cin, k, k)).astype(np.float32)


def _attention_seres18(rng, sd, name, c, ibn):
    mid = se_mid(c)
    sd[name + ".seblock.fc1.weight"] = rng.normal(0.0, np.sqrt(2.0 / c), (mid, c, 1, 1)).astype(np.float32)
    # the SE norm layer exists in the state_dict but is never applied
    # (SERes18_IBN.py:36 is commented out): LBN_1D for layers 1-3, BatchNorm1d for layer 4
    if ibn:
        h = mid // 2
        sd[name + ".seblock.bn.LN.weight"] = np.ones(h, np.float32)
        sd[name + ".seblock.bn.LN.bias"] = np.zeros(h, np.float32)
        _bn(rng, sd, name + ".seblock.bn.BN", mid - h)
    else:
        _bn(rng, sd, name + ".seblock.bn", mid)
    sd[name + ".seblock.fc2.weight"] = rng.normal(0.0, np.sqrt(2.0 / mid), (c, mid)).astype(np.float32)


def _attention_cares18(rng, sd, name, c, ibn):
    """TripletAttention (reid/backbones/triplet_attention.py:69-101): three AttentionGates, each conv7x7 (2 -> 1, no bias) + BN(1).
    CABasicBlock instantiates it in place of CABlock (CARes18.py:148)."""
    for gate in ("cw", "hc", "hw"):
        sd["%s.cablock.%s.conv.conv.weight" % (name, gate)] = rng.normal(0.0, 0.25, (1, 2, 7, 7)).astype(np.float32)
        _bn(rng, sd, "%s.cablock.%s.conv.bn" % (name, gate), 1)


def _attention_emares18(rng, sd, name, c, ibn):
    """EMA (reid/backbones/EMA_Res18.py:10-38), factor 32: GroupNorm, conv1x1 and conv3x3 over c / 32 channels, with biases."""
    cg = c // 32
    sd[name + ".emablock.gn.weight"] = rng.uniform(0.5, 1.5, cg).astype(np.float32)
    sd[name + ".emablock.gn.bias"] = rng.normal(0.0, 0.1, cg).astype(np.float32)
    sd[name + ".emablock.conv1x1.weight"] = rng.normal(0.0, np.sqrt(1.0 / cg), (cg, cg, 1, 1)).astype(np.float32)
    sd[name + ".emablock.conv1x1.bias"] = rng.normal(0.0, 0.1, cg).astype(np.float32)
    sd[name + ".emablock.conv3x3.weight"] = rng.normal(0.0, np.sqrt(1.0 / (9 * cg)), (cg, cg, 3, 3)).astype(np.float32)
    sd[name + ".emablock.conv3x3.bias"] = rng.normal(0.0, 0.1, cg).astype(np.float32)


_ATTENTION = {"seres18_ibn": _attention_seres18, "cares18_ibn": _attention_cares18, "emares18_ibn": _attention_emares18}


_SEQ_NAMES = ((".block_pre.conv1.", ".block_pre.0."), (".block_pre.bn1.", ".block_pre.1."), (".block_pre.conv2.", ".block_pre.3."),
              (".block_pre.bn2.", ".block_pre.4."), (".block_post.conv.", ".block_post.0."), (".block_post.bn.", ".block_post.1."))


def sibling_key(k, to_reference=True):
    """CABasicBlock / EMABasicBlock build block_pre of a downsample block as nn.Sequential(*children) (CARes18.py:141-142,
    EMA_Res18.py:69-70): its state_dict keys are positional (block_pre.0 = conv1, .1 = bn1, .3 = conv2, .4 = bn2; block_post.0 /
    .1 = the downsample conv / BN), where SEBasicBlock's are named.  Maps named <-> positional."""
    for named, pos in _SEQ_NAMES:
        a, b = (named, pos) if to_reference else (pos, named)
        if a in k:
            return k.replace(a, b)
    return k


def cares18_state_dict(seed=0, num_class=751, num_cams=6, gem_p=None):
    """numpy ``state_dict`` of CARes18_IBN (reid/backbones/CARes18.py:185-248): the SERse18_IBN skeleton with a
    TripletAttention per block instead of the SE block."""
    return seres18_state_dict(seed, num_class, num_cams, gem_p, arch="cares18_ibn")


def emares18_state_dict(seed=0, num_class=751, num_cams=6, gem_p=None):
    """numpy ``state_dict`` of EMARes18_IBN (reid/backbones/EMA_Res18.py:118-181): the same skeleton with an EMA block."""
    return seres18_state_dict(seed, num_class, num_cams, gem_p, arch="emares18_ibn")


def seres18_state_dict(seed=0, num_class=751, num_cams=6, gem_p=None, arch="seres18_ibn"):
    """numpy ``state_dict`` for SERse18_IBN (default ctor: gem pooling, se_lbn=True).

    Running stats, affine terms and (optionally) the GeM exponent are
    randomised so that a folding/ordering bug shows up in the outputs.
    ``arch`` selects the attention module of the sibling backbones (same conv skeleton, same key order otherwise).
    """
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    sd["cam_bias"] = rng.normal(0.0, 0.02, (num_cams, 512)).astype(np.float32)
    sd["conv0.weight"] = _conv(rng, 64, 3, 7)
    _bn(rng, sd, "bn0", 64)
    for name, c, ibn, ds, cin in SERES18_BLOCKS:
        pre = name + ".block_pre"
        sd[pre + ".conv1.weight"] = _conv(rng, c, cin, 3)
        if ibn:
            half = c // 2
            sd[pre + ".bn1.IN.weight"] = rng.uniform(0.5, 1.5, half).astype(np.float32)
            sd[pre + ".bn1.IN.bias"] = rng.normal(0.0, 0.1, half).astype(np.float32)
            _bn(rng, sd, pre + ".bn1.BN", c - half)
        else:
            _bn(rng, sd, pre + ".bn1", c)
        sd[pre + ".conv2.weight"] = _conv(rng, c, c, 3, gain=1.0)
        _bn(rng, sd, pre + ".bn2", c)
        if ds:
            sd[name + ".block_post.conv.weight"] = _conv(rng, c, cin, 1, gain=1.0)
            _bn(rng, sd, name + ".block_post.bn", c)
        _ATTENTION[arch](rng, sd, name, c, ibn)
    p = float(rng.uniform(2.5, 3.5)) if gem_p is None else float(gem_p)
    sd["avgpooling.p"] = np.asarray([p], dtype=np.float32)
    _bn(rng, sd, "bnneck", 512)
    sd["classifier.0.weight"] = rng.normal(0.0, 0.05, (num_class, 512)).astype(np.float32)
    if arch != "seres18_ibn":       # the siblings' downsample blocks carry positional keys (sibling_key)
        ds_blocks = tuple(b[0] for b in SERES18_BLOCKS if b[3])
        sd = OrderedDict((sibling_key(k) if k.startswith(ds_blocks) else k, v) for k, v in sd.items())
    return sd


def crops_u8(n, seed=0, h=256, w=128):
    """``uint8[n,h,w,3]`` uniform crops already at the extractor's (128, 256) size."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)


def smooth_crops_u8(n, seed=0, h=256, w=128):
    """Low-frequency crops: separable colour gradients + blobs, so that different
    crops give clearly different embeddings (rank tests need non-trivial gaps)."""
    rng = np.random.default_rng(seed)
    yy = np.linspace(0, 1, h, dtype=np.float32)[:, None, None]
    xx = np.linspace(0, 1, w, dtype=np.float32)[None, :, None]
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        a = rng.uniform(0, 255, (1, 1, 3)).astype(np.float32)
        b = rng.uniform(-200, 200, (1, 1, 3)).astype(np.float32)
        c = rng.uniform(-200, 200, (1, 1, 3)).astype(np.float32)
        f = rng.uniform(1, 6, 2)
        img = a + b * yy + c * xx + 60 * np.sin(2 * np.pi * f[0] * yy) * np.cos(2 * np.pi * f[1] * xx)
        img = img + rng.normal(0, 12, (h, w, 3)).astype(np.float32)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def ragged_crops_u8(n, seed=0):
    """Crops of different sizes (h in [40,400], w = h*U(0.3,0.5)), SURVEY §8d config 4."""
    rng = np.random.default_rng(seed)
    crops = []
    for _ in range(n):
        h = int(np.exp(rng.uniform(np.log(40), np.log(400))))
        w = max(8, int(h * rng.uniform(0.3, 0.5)))
        crops.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    return crops


def clustered_embeddings(nq, ng, d=512, n_ids=751, n_cams=6, seed=4, sigma=0.3):
    """L2-normalised embeddings around ``n_ids`` centroids + Market-like labels."""
    rng = np.random.default_rng(seed)
    cent = rng.normal(0, 1, (n_ids, d)).astype(np.float32)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    ql = rng.integers(1, n_ids, nq).astype(np.int64)
    gl = rng.integers(0, n_ids, ng).astype(np.int64)  # pid 0 = distractors
    qc = rng.integers(0, n_cams, nq).astype(np.int64)
    gc = rng.integers(0, n_cams, ng).astype(np.int64)
    qf = cent[ql] + sigma * rng.normal(0, 1, (nq, d)).astype(np.float32) / np.sqrt(d)
    gf = cent[gl] + sigma * rng.normal(0, 1, (ng, d)).astype(np.float32) / np.sqrt(d)
    qf = (qf / np.linalg.norm(qf, axis=1, keepdims=True)).astype(np.float32)
    gf = (gf / np.linalg.norm(gf, axis=1, keepdims=True)).astype(np.float32)
    return qf, ql, qc, gf, gl, gc


# ------------------------------------------------------------------------------------------------ Swin-T (v1)
SWIN_DIMS = (96, 192, 384, 768)
SWIN_LAYERS = (2, 2, 6, 2)
SWIN_HEADS = (3, 6, 12, 24)


def _lin(rng, sd, prefix, cout, cin, bias=True, gain=1.0):
    sd[prefix + ".weight"] = rng.normal(0.0, np.sqrt(gain / cin), (cout, cin)).astype(np.float32)
    if bias:
        sd[prefix + ".bias"] = rng.normal(0.0, 0.1, cout).astype(np.float32)


def _ln(rng, sd, prefix, c):
    sd[prefix + ".weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    sd[prefix + ".bias"] = rng.normal(0.0, 0.1, c).astype(np.float32)


def _swin_mask(ws, disp, upper_lower):
    """create_mask of reid/backbones/swin_transformer.py:95-108 (0 / -inf), restated."""
    m = np.zeros((ws * ws, ws * ws), np.float32)
    idx = np.arange(ws * ws)
    key = idx // ws if upper_lower else idx % ws
    hi = key >= ws - disp
    m[np.not_equal.outer(hi, hi)] = -np.inf
    return m


def swin_state_dict(seed=0, num_class=751, views=0):
    """numpy ``state_dict`` with exactly the keys/shapes of ``swin_t(version='v1').state_dict()``
    (reid/backbones/swin_transformer.py:339-395,508-513; 40.8 M parameters).  ``views`` > 0 adds the side-information table
    ``sfe.side_info_embedding`` [views,1,1,96] of a model built with camera / sequence (:285-293; views = camera * sequence, camera
    or sequence), drawn from a generator of its own so that the other tensors do not depend on it."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    sd["sfe.conv1.weight"] = rng.normal(0, np.sqrt(1.0 / 12), (12, 3, 2, 2)).astype(np.float32)
    sd["sfe.conv1.bias"] = rng.normal(0, 0.1, 12).astype(np.float32)
    sd["sfe.conv2.weight"] = rng.normal(0, np.sqrt(2.0 / 48), (48, 12, 2, 2)).astype(np.float32)
    sd["sfe.conv2.bias"] = rng.normal(0, 0.1, 48).astype(np.float32)
    sd["sfe.norm.instancenorm.weight"] = rng.uniform(0.5, 1.5, 6).astype(np.float32)
    sd["sfe.norm.instancenorm.bias"] = rng.normal(0, 0.1, 6).astype(np.float32)
    _bn(rng, sd, "sfe.norm.batchnorm", 6)
    _lin(rng, sd, "sfe.fc", 96, 48, gain=2.0)
    cin = 96
    for si, (c, nl, _h) in enumerate(zip(SWIN_DIMS, SWIN_LAYERS, SWIN_HEADS)):
        st = "stage%d" % (si + 1)
        down = 4 if si == 0 else 2
        _lin(rng, sd, st + ".patch_partition.linear", c, cin * down * down)   # stage1's is never used (patch_merge=False)
        for li in range(nl // 2):
            for bi in range(2):
                pre = "%s.layers.%d.%d" % (st, li, bi)
                _ln(rng, sd, pre + ".attention_block.fn.norm", c)
                if bi == 1:
                    sd[pre + ".attention_block.fn.fn.upper_lower_mask"] = _swin_mask(7, 3, True)
                    sd[pre + ".attention_block.fn.fn.left_right_mask"] = _swin_mask(7, 3, False)
                sd[pre + ".attention_block.fn.fn.pos_embedding"] = rng.normal(0, 0.5, (13, 13)).astype(np.float32)
                _lin(rng, sd, pre + ".attention_block.fn.fn.to_qkv", 3 * c, c, bias=False)
                _lin(rng, sd, pre + ".attention_block.fn.fn.to_out", c, c, gain=0.5)
                _lin(rng, sd, pre + ".attention_block.fn.fn.post_proj", c, c, gain=0.5)
                _ln(rng, sd, pre + ".mlp_block.fn.norm", c)
                _lin(rng, sd, pre + ".mlp_block.fn.fn.net.0", 4 * c, c, gain=2.0)
                _lin(rng, sd, pre + ".mlp_block.fn.fn.net.3", c, 4 * c, gain=0.5)
        cin = c
    _ln(rng, sd, "norm", 96)
    _bn(rng, sd, "bottleneck", 96)
    sd["mlp_head.0.weight"] = rng.normal(0, 0.05, (num_class, 96)).astype(np.float32)
    sd["img_channel_align.weight"] = rng.normal(0, np.sqrt(0.5 / (96 * 64)), (768, 96, 8, 8)).astype(np.float32)
    sd["img_channel_align.bias"] = rng.normal(0, 0.1, 768).astype(np.float32)
    for name, ci, co in (("stage4_channel_align", 768, 384), ("stage3_channel_align", 384, 192), ("stage2_channel_align", 192, 96)):
        sd[name + ".weight"] = rng.normal(0, np.sqrt(0.5 / (ci * 4)), (ci, co, 4, 4)).astype(np.float32)
        sd[name + ".bias"] = rng.normal(0, 0.1, co).astype(np.float32)
    sd["avgpool.p"] = np.asarray([float(rng.uniform(2.5, 3.5))], np.float32)
    if views > 0:
        sd["sfe.side_info_embedding"] = np.random.default_rng(seed + 7919).normal(0.0, 0.5, (views, 1, 1, 96)).astype(np.float32)
        sd.move_to_end("sfe.side_info_embedding", last=False)   # a Parameter of sfe: first in the reference's key order
    return sd


def images_f32(n, seed=0, h=224, w=224):
    """Normalised float images [n,3,h,w] in [-1,1] with low-frequency structure (Swin needs 224x224)."""
    rng = np.random.default_rng(seed)
    yy = np.linspace(0, 1, h, dtype=np.float32)[None, None, :, None]
    xx = np.linspace(0, 1, w, dtype=np.float32)[None, None, None, :]
    a = rng.uniform(-1, 1, (n, 3, 1, 1)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, 3, 1, 1)).astype(np.float32)
    f = rng.uniform(1, 5, (n, 3, 1, 1)).astype(np.float32)
    x = 0.5 * a * yy + 0.5 * b * xx + 0.3 * np.sin(2 * np.pi * f * yy) * np.cos(2 * np.pi * f * xx)
    x = x + rng.normal(0, 0.15, (n, 3, h, w)).astype(np.float32)
    return np.clip(x, -1, 1).astype(np.float32)


def noise_images_f32(n, seed=0, h=224, w=224):
    """Normalised float images [n,3,h,w], every pixel uniform in [-1,1): the Swin counterpart of ``crops_u8`` (embeddings of noise
    are nearly parallel, so the reference's top-2 gaps are tiny - the hard set of tests/golden/swin_config.npz)."""
    rng = np.random.default_rng(seed)
    return rng.uniform(-1.0, 1.0, (n, 3, h, w)).astype(np.float32)


# ------------------------------------------------------------------------------------------------ evaluation-harness problem
def _id_pattern(rng, h, w):
    yy = np.linspace(0, 1, h, dtype=np.float32)[None, :, None]
    xx = np.linspace(0, 1, w, dtype=np.float32)[None, None, :]
    a = rng.uniform(-1, 1, (3, 1, 1)).astype(np.float32)
    b = rng.uniform(-1.5, 1.5, (3, 1, 1)).astype(np.float32)
    c = rng.uniform(-1.5, 1.5, (3, 1, 1)).astype(np.float32)
    f = rng.uniform(1, 5, 2)
    ph = rng.uniform(0, 6.28, 2)
    wave = np.sin(2 * np.pi * f[0] * yy + ph[0]) * np.cos(2 * np.pi * f[1] * xx + ph[1])
    return (a + b * (yy - 0.5) + c * (xx - 0.5) + 0.6 * wave).astype(np.float32)


def identity_images_f32(labels, cams, seed, noise=0.2, cam_shift=0.4, h=256, w=128):
    """Normalised float images [n,3,h,w] with identity AND camera structure: image = gain * pattern[label] + cam_shift *
    pattern[camera] + noise (white + one low-frequency pattern).  The identity / camera patterns come from a fixed stream, so
    a gallery and a query set generated with different ``seed`` share them.  Stands in for a dataset the evaluation
    script (reid/image_reid_inference.py) would load: retrieval is non-trivial and the per-camera offset is what
    ``diminish_camera_bias`` removes."""
    labels, cams = np.asarray(labels), np.asarray(cams)
    brng = np.random.default_rng(7)
    base = np.stack([_id_pattern(brng, h, w) for _ in range(max(32, int(labels.max()) + 1))])
    camp = np.stack([_id_pattern(brng, h, w) for _ in range(max(8, int(cams.max()) + 1))])
    rng = np.random.default_rng(seed)
    out = np.empty((len(labels), 3, h, w), np.float32)
    for i, (l, c) in enumerate(zip(labels, cams)):
        g = np.float32(rng.uniform(0.8, 1.2))
        out[i] = (g * base[l] + np.float32(cam_shift) * camp[c] + np.float32(noise) * rng.normal(0, 1, (3, h, w)).astype(np.float32)
                  + np.float32(noise) * _id_pattern(rng, h, w))
    return out


def e2e_problem(seed=100, n_ids=24, n_cams=4, n_gallery=300, n_query=60, n_seqs=5):
    """The seeded retrieval problem of the end-to-end harness test (tests/golden/e2e.npz): labels, cameras, sequence ids and
    images of a gallery and a query set.  Label 0 of the gallery plays Market's distractors (queries draw from 1..)."""
    rng = np.random.default_rng(seed)
    gl = rng.integers(0, n_ids, n_gallery).astype(np.int64)
    gc = rng.integers(0, n_cams, n_gallery).astype(np.int64)
    ql = rng.integers(1, n_ids, n_query).astype(np.int64)
    qc = rng.integers(0, n_cams, n_query).astype(np.int64)
    gs = rng.integers(0, n_seqs, n_gallery).astype(np.int64)
    qs = rng.integers(0, n_seqs, n_query).astype(np.int64)
    return {"gl": gl, "gc": gc, "gs": gs, "ql": ql, "qc": qc, "qs": qs,
            "g_img": identity_images_f32(gl, gc, seed + 1), "q_img": identity_images_f32(ql, qc, seed + 2)}


def renorm_state_dict(sd):
    """The ``state_dict`` layout of ``seres18_ibn(renorm=True)`` (checkpoints trained with ``--renorm``,
    reid/image_reid_inference.py:154,180-181) from a plain one: every BatchNorm2d the constructor swaps for
    ``BatchRenormalization2D`` (SERes18_IBN.py:102-113,203-204: bn0, each block's bn1 - or the BN half of its IBN -, bn2 and
    the shortcut's bn) carries ``gamma / beta / running_avg_mean / running_avg_var`` shaped [1,C,1,1] plus the scalars
    ``num_tracked_batch, r_max, d_max`` (batchrenorm.py:26-40); the unused SE norm of layers 1-3 becomes a
    ``BatchRenormalization1D`` ([1,C]).  bnneck and layer 4's SE norm stay BatchNorm1d."""
    out = OrderedDict()
    ren = {"weight": "gamma", "bias": "beta", "running_mean": "running_avg_mean", "running_var": "running_avg_var"}
    for k, v in sd.items():
        prefix, _, leaf = k.rpartition(".")
        is_bn = (prefix + ".running_mean") in sd
        plain = prefix == "bnneck" or (prefix.endswith("seblock.bn") and is_bn)     # BatchNorm1d in both layouts
        if not is_bn or plain:
            out[k] = v
            continue
        shape = (1, -1) if prefix.endswith("seblock.bn.BN") else (1, -1, 1, 1)
        if leaf in ren:
            out[prefix + "." + ren[leaf]] = np.asarray(v, np.float32).reshape(shape)
        elif leaf == "num_batches_tracked":
            out[prefix + ".num_tracked_batch"] = np.asarray(0, dtype=np.int64)
            out[prefix + ".r_max"] = np.asarray(1.0, dtype=np.float32)
            out[prefix + ".d_max"] = np.asarray(0.0, dtype=np.float32)
    return out


def tracking_stream(frames=600, seed=3, pool_size=256, max_dets=80):
    """Stand-in for the MOT16-02 detection dump of BASELINE configs[3] (not in the container):
    ``frames`` frames, detections per frame ~ Poisson(30) clipped to [1, max_dets] (MOT16-02: 17 833 boxes over 600 frames),
    crops drawn from a pool of ragged crops (h log-uniform in [40, 400], w = h * U(0.3, 0.5)), tlwh boxes.
    Returns (counts int[frames], pool list of uint8[h,w,3], boxes float64[max_dets,4], crops_of(f) -> the frame's crop list)."""
    rng = np.random.default_rng(seed)
    counts = np.clip(rng.poisson(30, frames), 1, max_dets)
    pool = ragged_crops_u8(pool_size, seed=seed)
    boxes = rng.uniform(0, 500, (max_dets, 4))
    boxes[:, 2:] = rng.uniform(20, 120, (max_dets, 2))

    def crops_of(f):
        return [pool[(f * 7 + i) % pool_size] for i in range(int(counts[f]))]
    return counts, pool, boxes, crops_of
