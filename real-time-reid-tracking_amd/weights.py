"""state_dict -> packed fp32 blob + manifest for libreid_hip.so.

Weight-ingest contract of the reference (checkpoint row):
  * checkpoints are ``torch.save(model.state_dict())`` of a DataParallel model, so keys may carry a
    ``module.`` prefix (image_reid_train.py:111,635); ``{'state_dict': ...}`` wrappers are accepted
    (reid_model_factory.py:172-175);
  * ``Extractor`` loads with ``strict=False`` (feature_extractor.py:18-19): unknown keys are ignored;
  * tensors the forward pass never reads are dropped: ``*.seblock.bn.*`` (SERes18_IBN.py:36 commented out),
    ``num_batches_tracked``; ``cam_bias`` (read only when ``cam`` is passed, SERes18_IBN.py:269-270) and Swin's
    ``sfe.side_info_embedding`` (swin_transformer.py:301-302) travel as optional tables with their constructor coefficient;
  * ``--renorm`` checkpoints hold BatchRenormalization2D layers (gamma/beta/running_avg_*), whose eval
    math is plain BN (batchrenorm.py:93-95).

Packing for the kernels:
  * eval BatchNorm is folded to per-channel (scale, shift) in float64 and rounded once;
  * conv weights go from [Cout][Cin][R][S] to [Cout][R][S][Cin] (K contiguous, matching the NHWC im2col order);
  * the 7x7 stem is stored as [64][8][24]: 7 kernel rows x (7 taps x 3 channels = 21) padded to 8 x 24 = 192.
"""
from collections import OrderedDict

import numpy as np

from . import synth

BN_EPS = 1e-5
_BLK_SHORT = {"basicBlock11": "b11", "basicBlock12": "b12", "basicBlock21": "b21", "basicBlock22": "b22",
              "basicBlock31": "b31", "basicBlock32": "b32", "basicBlock41": "b41", "basicBlock42": "b42"}


def _np(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def normalize_state_dict(state_dict):
    """Unwraps {'state_dict': ...}, strips 'module.', converts to numpy; the positional keys of the sibling backbones' downsample
    blocks (block_pre.0 / .1 / .3 / .4, block_post.0 / .1: CARes18.py:141-142) become the named ones of SERse18_IBN."""
    if isinstance(state_dict, dict) and "state_dict" in state_dict and not hasattr(state_dict["state_dict"], "shape"):
        state_dict = state_dict["state_dict"]
    out = OrderedDict()
    for k, v in state_dict.items():
        if k.startswith("module."):
            k = k[7:]
        if k.startswith("basicBlock"):
            k = synth.sibling_key(k, to_reference=False)
        out[k] = _np(v)
    return _renorm_to_plain(out)


_RENORM_KEYS = {"gamma": "weight", "beta": "bias", "running_avg_mean": "running_mean", "running_avg_var": "running_var",
                "num_tracked_batch": "num_batches_tracked"}


def _renorm_to_plain(sd):
    """`--renorm` checkpoints (seres18_ibn(renorm=True), SERes18_IBN.py:102-113,203-204): BatchRenormalization layers hold
    gamma / beta / running_avg_mean / running_avg_var shaped [1,C,1,1] (or [1,C]) plus num_tracked_batch, r_max, d_max
    (batchrenorm.py:26-40).  In eval mode they ARE BatchNorm (batchrenorm.py:93-95), so the keys are renamed to BatchNorm's and
    flattened; r_max / d_max only steer training and are dropped."""
    if not any(k.endswith(".running_avg_mean") for k in sd):
        return sd
    out = OrderedDict()
    for k, v in sd.items():
        prefix, _, leaf = k.rpartition(".")
        if (prefix + ".running_avg_mean") not in sd:
            out[k] = v
        elif leaf in _RENORM_KEYS:
            out[prefix + "." + _RENORM_KEYS[leaf]] = v.reshape(-1) if leaf != "num_tracked_batch" else v
    return out


def fold_bn(sd, prefix):
    """Eval-mode BN (or BatchRenormalization2D) -> (scale, shift) float32."""
    if prefix + ".weight" in sd:
        g, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
        m, v = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    elif prefix + ".gamma" in sd:
        g, b = sd[prefix + ".gamma"], sd[prefix + ".beta"]
        m, v = sd[prefix + ".running_avg_mean"], sd[prefix + ".running_avg_var"]
    else:
        raise KeyError("no BatchNorm parameters under '%s'" % prefix)
    g, b, m, v = (np.asarray(t, np.float64).reshape(-1) for t in (g, b, m, v))
    scale = g / np.sqrt(v + BN_EPS)
    return scale.astype(np.float32), (b - m * scale).astype(np.float32)


def conv_krsc(w):
    """[Cout][Cin][R][S] -> [Cout][R*S*Cin] float32."""
    w = np.asarray(w, np.float32)
    return np.ascontiguousarray(w.transpose(0, 2, 3, 1)).reshape(w.shape[0], -1)


def stem_pack(w):
    """[64][3][7][7] -> [64][8][24]: k = r*24 + s*3 + c, zero padded."""
    w = np.asarray(w, np.float32)
    out = np.zeros((w.shape[0], 8, 24), np.float32)
    out[:, :7, :21] = w.transpose(0, 2, 3, 1).reshape(w.shape[0], 7, 21)
    return out.reshape(w.shape[0], 192)


class Packer:
    def __init__(self):
        self.parts, self.lines, self.off = [], [], 0

    def add(self, name, arr):
        arr = np.ascontiguousarray(np.asarray(arr, np.float32).reshape(-1))
        self.lines.append("%s %d %d" % (name, self.off, arr.size))
        pad = (-arr.size) % 4            # keep every tensor 16-byte aligned
        self.parts.append(arr)
        if pad:
            self.parts.append(np.zeros(pad, np.float32))
        self.off += arr.size + pad

    def finish(self):
        return np.concatenate(self.parts).astype(np.float32), "\n".join(self.lines) + "\n"


def pack_seres18(state_dict, cam_factor=-1.0):
    """Returns (blob float32[n], manifest str, info dict) for reid_seres18_load.  ``cam_factor``: the constructor argument of
    SERse18_IBN (SERes18_IBN.py:198, not part of the state_dict) that scales the camera-bias term."""
    sd = normalize_state_dict(state_dict)
    required = ["conv0.weight", "basicBlock11.block_pre.conv1.weight", "bnneck.running_mean"]
    for k in required:
        if k not in sd:
            raise KeyError("state_dict is not a SERse18_IBN checkpoint: missing '%s'" % k)
    arch = ("cares18_ibn" if any(".cablock." in k for k in sd) else
            "emares18_ibn" if any(".emablock." in k for k in sd) else "seres18_ibn")
    pk = Packer()
    pk.add("stem.w", stem_pack(sd["conv0.weight"]))
    s, b = fold_bn(sd, "bn0")
    pk.add("stem.scale", s)
    pk.add("stem.shift", b)
    for name, c, ibn, ds, cin in synth.SERES18_BLOCKS:
        sh, pre = _BLK_SHORT[name], name + ".block_pre"
        pk.add(sh + ".conv1.w", conv_krsc(sd[pre + ".conv1.weight"]))
        if ibn:
            pk.add(sh + ".n1.in_gamma", sd[pre + ".bn1.IN.weight"])
            pk.add(sh + ".n1.in_beta", sd[pre + ".bn1.IN.bias"])
            s, b = fold_bn(sd, pre + ".bn1.BN")
        else:
            s, b = fold_bn(sd, pre + ".bn1")
        pk.add(sh + ".n1.bn_scale", s)
        pk.add(sh + ".n1.bn_shift", b)
        pk.add(sh + ".conv2.w", conv_krsc(sd[pre + ".conv2.weight"]))
        s, b = fold_bn(sd, pre + ".bn2")
        pk.add(sh + ".bn2.scale", s)
        pk.add(sh + ".bn2.shift", b)
        if ds:
            pk.add(sh + ".ds.w", conv_krsc(sd[name + ".block_post.conv.weight"]))
            s, b = fold_bn(sd, name + ".block_post.bn")
            pk.add(sh + ".ds.scale", s)
            pk.add(sh + ".ds.shift", b)
        if arch == "cares18_ibn":
            # TripletAttention (triplet_attention.py:69-101): per gate the 7x7 conv weight [2][7][7] (std, mean planes) + folded BN(1)
            gates = []
            for gate in ("cw", "hc", "hw"):
                gp = "%s.cablock.%s.conv" % (name, gate)
                gs, gb = fold_bn(sd, gp + ".bn")
                gates.append(np.concatenate([np.asarray(sd[gp + ".conv.weight"], np.float32).reshape(98), gs, gb]))
            pk.add(sh + ".ta", np.concatenate(gates))
        elif arch == "emares18_ibn":
            # EMA (EMA_Res18.py:10-22): conv1x1 w, b | conv3x3 w [co][ci][3][3], b | GroupNorm weight, bias
            ep = name + ".emablock"
            pk.add(sh + ".ema", np.concatenate([np.asarray(sd[ep + k], np.float32).reshape(-1) for k in
                                                (".conv1x1.weight", ".conv1x1.bias", ".conv3x3.weight", ".conv3x3.bias",
                                                 ".gn.weight", ".gn.bias")]))
        else:
            mid = synth.se_mid(c)
            pk.add(sh + ".se.w1", np.asarray(sd[name + ".seblock.fc1.weight"], np.float32).reshape(mid, c))
            # fc2 stored transposed [mid][C]: the SE kernel reads it with consecutive threads on consecutive channels
            pk.add(sh + ".se.w2t", np.ascontiguousarray(np.asarray(sd[name + ".seblock.fc2.weight"], np.float32).reshape(c, mid).T))
    p = sd.get("avgpooling.p", np.asarray([3.0], np.float32))   # GeM init p=3 (attention_pooling.py:52)
    pk.add("gem.p", np.asarray(p, np.float32).reshape(1))
    s, b = fold_bn(sd, "bnneck")
    pk.add("neck.scale", s)
    pk.add("neck.shift", b)
    num_class = 0
    if "classifier.0.weight" in sd:
        w = np.asarray(sd["classifier.0.weight"], np.float32)
        num_class = w.shape[0]
        pk.add("cls.w", w)
    if "cam_bias" in sd and np.asarray(sd["cam_bias"]).ndim == 2 and np.asarray(sd["cam_bias"]).shape[1] == 512:
        pk.add("cam.bias", np.asarray(sd["cam_bias"], np.float32))
        pk.add("cam.factor", np.asarray([cam_factor], np.float32))
    blob, manifest = pk.finish()
    return blob, manifest, {"arch": arch, "embed_dim": 512, "num_class": num_class}


# ------------------------------------------------------------------------------------------------ Swin-T (v1)
def _convt_parity(w):
    """ConvTranspose2d(4, 2, 1) weight [Cin][Cout][4][4] -> [py][px][Cout][r][s][Cin]: the four output-parity 2x2
    convolutions.  Output row 2j+py takes input rows j-1+r (py=0: kernel rows 3,1) or j+r (py=1: kernel rows 2,0)."""
    w = np.asarray(w, np.float32)
    ci, co = w.shape[:2]
    taps = ((3, 1), (2, 0))
    out = np.empty((2, 2, co, 2, 2, ci), np.float32)
    for py in range(2):
        for px in range(2):
            for r in range(2):
                for s in range(2):
                    out[py, px, :, r, s, :] = w[:, :, taps[py][r], taps[px][s]].T
    return out


def pack_swin(state_dict, side_info_coeff=1.5):
    """state_dict of swin_t(version='v1') (reid/backbones/swin_transformer.py) -> (blob, manifest, info) for reid_swin_load.
    ``side_info_coeff``: ShadowFeatureExtraction's constructor argument (:279) that scales the side-information embedding.
    Dropped: stage1.patch_partition (never applied, patch_merge=False :357-359), the constant shift masks (recomputed in
    the attention kernel), num_batches_tracked."""
    sd = normalize_state_dict(state_dict)
    if "sfe.conv1.weight" not in sd or "stage4.layers.0.1.attention_block.fn.fn.to_qkv.weight" not in sd:
        raise KeyError("state_dict is not a swin_t (v1) checkpoint")
    f = lambda k: np.asarray(sd[k], np.float32)
    pk = Packer()
    pk.add("sfe.conv1.w", f("sfe.conv1.weight").transpose(0, 2, 3, 1))       # [12][(kh,kw,c)]
    pk.add("sfe.conv1.b", f("sfe.conv1.bias"))
    pk.add("sfe.in_gamma", f("sfe.norm.instancenorm.weight"))
    pk.add("sfe.in_beta", f("sfe.norm.instancenorm.bias"))
    s, b = fold_bn(sd, "sfe.norm.batchnorm")
    pk.add("sfe.bn_scale", s)
    pk.add("sfe.bn_shift", b)
    pk.add("sfe.conv2.w", f("sfe.conv2.weight").transpose(0, 2, 3, 1))       # [48][(kh,kw,c)]
    pk.add("sfe.conv2.b", f("sfe.conv2.bias"))
    pk.add("sfe.fc.w", f("sfe.fc.weight"))
    pk.add("sfe.fc.b", f("sfe.fc.bias"))
    cin = 96
    for si, (c, nl) in enumerate(zip(synth.SWIN_DIMS, synth.SWIN_LAYERS)):
        st, out = "stage%d" % (si + 1), "s%d" % (si + 1)
        if si > 0:
            w = f(st + ".patch_partition.linear.weight")                     # [C][(c_in, kh, kw)] (nn.Unfold order)
            pk.add(out + ".merge.w", w.reshape(c, cin, 2, 2).transpose(0, 2, 3, 1))   # -> [C][(kh, kw, c_in)]
            pk.add(out + ".merge.b", f(st + ".patch_partition.linear.bias"))
        j = 0
        for li in range(nl // 2):
            for bi in range(2):
                pre = "%s.layers.%d.%d" % (st, li, bi)
                a, m, o = pre + ".attention_block.fn", pre + ".mlp_block.fn", "%s.b%d" % (out, j)
                pk.add(o + ".ln1.g", f(a + ".norm.weight"))
                pk.add(o + ".ln1.b", f(a + ".norm.bias"))
                pk.add(o + ".qkv.w", f(a + ".fn.to_qkv.weight"))
                pk.add(o + ".pos", f(a + ".fn.pos_embedding").reshape(169))
                pk.add(o + ".out.w", f(a + ".fn.to_out.weight"))
                pk.add(o + ".out.b", f(a + ".fn.to_out.bias"))
                pk.add(o + ".post.w", f(a + ".fn.post_proj.weight"))
                pk.add(o + ".post.b", f(a + ".fn.post_proj.bias"))
                pk.add(o + ".ln2.g", f(m + ".norm.weight"))
                pk.add(o + ".ln2.b", f(m + ".norm.bias"))
                pk.add(o + ".fc1.w", f(m + ".fn.net.0.weight"))
                pk.add(o + ".fc1.b", f(m + ".fn.net.0.bias"))
                pk.add(o + ".fc2.w", f(m + ".fn.net.3.weight"))
                pk.add(o + ".fc2.b", f(m + ".fn.net.3.bias"))
                j += 1
        cin = c
    pk.add("align.img.w", f("img_channel_align.weight").transpose(0, 2, 3, 1))   # [768][8][8][96]
    pk.add("align.img.b", f("img_channel_align.bias"))
    for t, name in enumerate(("stage4_channel_align", "stage3_channel_align", "stage2_channel_align")):
        pk.add("align.t%d.w" % t, _convt_parity(sd[name + ".weight"]))
        pk.add("align.t%d.b" % t, f(name + ".bias"))
    pk.add("tail.ln.g", f("norm.weight"))
    pk.add("tail.ln.b", f("norm.bias"))
    pk.add("tail.p", np.asarray(sd.get("avgpool.p", [3.0]), np.float32).reshape(1))
    s, b = fold_bn(sd, "bottleneck")
    pk.add("tail.bn_scale", s)
    pk.add("tail.bn_shift", b)
    num_class = 0
    if "mlp_head.0.weight" in sd:
        w = f("mlp_head.0.weight")
        num_class = w.shape[0]
        pk.add("cls.w", w)
    if "sfe.side_info_embedding" in sd:
        pk.add("sfe.side", f("sfe.side_info_embedding").reshape(-1, 96))
        pk.add("sfe.side_coeff", np.asarray([side_info_coeff], np.float32))
    blob, manifest = pk.finish()
    return blob, manifest, {"arch": "swin_transformer", "embed_dim": 96, "num_class": num_class}
