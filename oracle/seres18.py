"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the reference's ResNet18-IBN-SE ("ResNet18-SE") eval-mode
forward, written from the reference's source as plain functional torch-CPU ops
on a ``state_dict`` - no ``nn.Module`` from the reference, no third-party hub
model.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file.

Pinned against the reference's own classes (run here with stubbed
``torchvision`` / ``torch.hub``) by ``oracle/gen_golden.py`` ->
``tests/golden/seres18_*.npz`` and ``tests/test_oracle_golden.py``.

Follows (all paths under /root/reference):
  reid/backbones/SERes18_IBN.py:250-276   SERse18_IBN.forward (stem has NO ReLU, :253)
  reid/backbones/SERes18_IBN.py:96-128    SEBasicBlock (double residual for blocks w/o downsample)
  reid/backbones/SERes18_IBN.py:88-93     IBN.forward (IN on first half, eval-BN on second half)
  reid/backbones/SERes18_IBN.py:32-41     SEBlock.forward (its norm layer is commented out, :36)
  reid/backbones/attention_pooling.py:58-60  GeM
  [external] XingangPan/IBN-Net resnet18_ibn_a BasicBlock_IBN (conv1,bn1,relu,conv2,bn2,+res,relu)
Sibling backbones on the same skeleton (``arch``), pinned the same way (tests/golden/siblings.npz):
  reid/backbones/CARes18.py:102-162,185-281   CARes18_IBN: block_pre -> TripletAttention -> + shortcut -> ReLU
  reid/backbones/triplet_attention.py:46-101  ZPool (unbiased std, mean) -> conv7x7 -> BN -> sigmoid, three orientations
  reid/backbones/EMA_Res18.py:10-38,41-86     EMARes18_IBN: block_pre -> EMA(factor 32) -> + shortcut -> ReLU
Eval-mode semantics everywhere: BN uses running statistics.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
IN_EPS = 1e-5
GEM_EPS = 1e-6

BLOCKS = [  # name, channels, ibn, downsample, stride of conv1
    ("basicBlock11", 64, True, False, 1),
    ("basicBlock12", 64, True, False, 1),
    ("basicBlock21", 128, True, True, 2),
    ("basicBlock22", 128, True, False, 1),
    ("basicBlock31", 256, True, True, 2),
    ("basicBlock32", 256, True, False, 1),
    ("basicBlock41", 512, False, True, 1),   # last stride forced to 1 (SERes18_IBN.py:99-101,223)
    ("basicBlock42", 512, False, False, 1),
]


_POSITIONAL = ((".block_pre.conv1.", ".block_pre.0."), (".block_pre.bn1.", ".block_pre.1."), (".block_pre.conv2.", ".block_pre.3."),
               (".block_pre.bn2.", ".block_pre.4."), (".block_post.conv.", ".block_post.0."), (".block_post.bn.", ".block_post.1."))


def _t(sd, k):
    if k not in sd:      # CABasicBlock / EMABasicBlock: block_pre of a downsample block is a positional nn.Sequential
        for named, pos in _POSITIONAL:   # (CARes18.py:141-142, EMA_Res18.py:69-70)
            if named in k:
                k = k.replace(named, pos)
                break
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _bn(sd, prefix, x):
    if prefix + ".gamma" in sd:      # --renorm checkpoint: BatchRenormalization2D, eval branch (batchrenorm.py:93-95)
        mean, var = _t(sd, prefix + ".running_avg_mean"), _t(sd, prefix + ".running_avg_var")
        return _t(sd, prefix + ".gamma") * ((x - mean) / torch.sqrt(var + BN_EPS)) + _t(sd, prefix + ".beta")
    return F.batch_norm(x, _t(sd, prefix + ".running_mean"), _t(sd, prefix + ".running_var"),
                        _t(sd, prefix + ".weight"), _t(sd, prefix + ".bias"), False, 0.0, BN_EPS)


def _ibn(sd, prefix, x):
    half = x.shape[1] // 2
    a = F.instance_norm(x[:, :half].contiguous(), None, None, _t(sd, prefix + ".IN.weight"),
                        _t(sd, prefix + ".IN.bias"), True, 0.0, IN_EPS)
    b = _bn(sd, prefix + ".BN", x[:, half:].contiguous())
    return torch.cat((a, b), 1)


def _se(sd, prefix, y):
    n, c = y.shape[:2]
    pooled = y.mean(dim=(2, 3))                               # AdaptiveAvgPool2d(1)
    h = F.relu(pooled @ _t(sd, prefix + ".fc1.weight").reshape(-1, c).t())   # 1x1 conv, no bias
    return torch.sigmoid(h @ _t(sd, prefix + ".fc2.weight").t()).reshape(n, c, 1, 1)


def _gate(sd, prefix, x):
    """AttentionGate (triplet_attention.py:55-66): ZPool over dim 1 -> conv 7x7 (2 -> 1) -> BN(1) -> sigmoid -> x * scale."""
    z = torch.cat((torch.std(x, 1).unsqueeze(1), torch.mean(x, 1).unsqueeze(1)), dim=1)
    g = _bn(sd, prefix + ".conv.bn", F.conv2d(z, _t(sd, prefix + ".conv.conv.weight"), None, 1, 3))
    return x * torch.sigmoid(g)


def _triplet(sd, prefix, y):
    """TripletAttention.forward (triplet_attention.py:88-101), no_spatial=False."""
    o1 = _gate(sd, prefix + ".cw", y.permute(0, 2, 1, 3).contiguous()).permute(0, 2, 1, 3).contiguous()
    o2 = _gate(sd, prefix + ".hc", y.permute(0, 3, 2, 1).contiguous()).permute(0, 3, 2, 1).contiguous()
    return 1 / 3 * (_gate(sd, prefix + ".hw", y) + o1 + o2)


def _ema(sd, prefix, y, groups=32):
    """EMA.forward (EMA_Res18.py:23-38)."""
    b, c, h, w = y.shape
    cg = c // groups
    gx = y.reshape(b * groups, cg, h, w)
    x_h = gx.mean(dim=3, keepdim=True)                                  # pool_h: (None, 1)
    x_w = gx.mean(dim=2, keepdim=True).permute(0, 1, 3, 2)              # pool_w: (1, None), permuted
    hw = F.conv2d(torch.cat([x_h, x_w], dim=2), _t(sd, prefix + ".conv1x1.weight"), _t(sd, prefix + ".conv1x1.bias"))
    x_h, x_w = torch.split(hw, [h, w], dim=2)
    x1 = F.group_norm(gx * x_h.sigmoid() * x_w.permute(0, 1, 3, 2).sigmoid(), cg, _t(sd, prefix + ".gn.weight"),
                      _t(sd, prefix + ".gn.bias"), 1e-5)
    x2 = F.conv2d(gx, _t(sd, prefix + ".conv3x3.weight"), _t(sd, prefix + ".conv3x3.bias"), 1, 1)
    x11 = torch.softmax(x1.mean(dim=(2, 3)).reshape(b * groups, -1, 1).permute(0, 2, 1), -1)
    x12 = x2.reshape(b * groups, cg, -1)
    x21 = torch.softmax(x2.mean(dim=(2, 3)).reshape(b * groups, -1, 1).permute(0, 2, 1), -1)
    x22 = x1.reshape(b * groups, cg, -1)
    weights = (torch.matmul(x11, x12) + torch.matmul(x21, x22)).reshape(b * groups, 1, h, w)
    return (gx * weights.sigmoid()).reshape(b, c, h, w)


def _block(sd, name, ibn, ds, stride, x, taps=None, arch="seres18_ibn"):
    pre = name + ".block_pre"
    c1 = F.conv2d(x, _t(sd, pre + ".conv1.weight"), None, stride, 1)
    h = F.relu(_ibn(sd, pre + ".bn1", c1) if ibn else _bn(sd, pre + ".bn1", c1))
    y = _bn(sd, pre + ".bn2", F.conv2d(h, _t(sd, pre + ".conv2.weight"), None, 1, 1))
    if ds:
        sc = _bn(sd, name + ".block_post.bn", F.conv2d(x, _t(sd, name + ".block_post.conv.weight"), None, stride, 0))
    else:
        y = F.relu(y + x)          # BasicBlock_IBN's own residual + relu (block_pre is the whole block)
        sc = x
    if arch == "cares18_ibn":       # CABasicBlock.forward, CARes18.py:150-157
        out = F.relu(_triplet(sd, name + ".cablock", y) + sc)
    elif arch == "emares18_ibn":    # EMABasicBlock.forward, EMA_Res18.py:79-86
        out = F.relu(_ema(sd, name + ".emablock", y) + sc)
    else:
        s = _se(sd, name + ".seblock", y)
        out = F.relu(s * y + sc)
        if taps is not None:
            taps[name + ".se"] = s.reshape(s.shape[0], -1)
    if taps is not None:
        taps[name + ".conv1"] = c1
        taps[name + ".y"] = y
        taps[name] = out
    return out


def preprocess_u8(crops_u8):
    """uint8[N,H,W,3] crops already at 128x256 -> float32[N,3,H,W], (x/255-0.5)/0.5.
    feature_extractor.py:40-46 (cv2.resize to the same size is the identity)."""
    x = torch.from_numpy(np.ascontiguousarray(crops_u8)).to(torch.float32) / 255.0
    x = (x - 0.5) / 0.5
    return x.permute(0, 3, 1, 2).contiguous()


def forward(sd, x, taps=None, arch="seres18_ibn", cam=None, cam_factor=-1.0):
    """x: float32[N,3,256,128] NCHW (normalised).  Returns (emb[N,512], logits[N,num_class]).  ``arch``: seres18_ibn (default),
    cares18_ibn or emares18_ibn.  ``cam`` (camera index per image) adds cam_factor * cam_bias[cam] to the BNNeck output before the
    classifier (SERes18_IBN.py:269-270; cam_factor is the constructor's, default -1, :198)."""
    with torch.no_grad():
        x = F.conv2d(x, _t(sd, "conv0.weight"), None, 2, 3)
        x = _bn(sd, "bn0", x)                                  # no ReLU (SERes18_IBN.py:253)
        if taps is not None:
            taps["stem"] = x
        x = F.max_pool2d(x, 3, 2, 1)
        if taps is not None:
            taps["pool0"] = x
        for name, _c, ibn, ds, stride in BLOCKS:
            x = _block(sd, name, ibn, ds, stride, x, taps, arch)
        p = _t(sd, "avgpooling.p")
        feat = x.clamp(min=GEM_EPS).pow(p).mean(dim=(2, 3)).pow(1.0 / p)
        if taps is not None:
            taps["gem"] = feat
        emb = F.batch_norm(feat, _t(sd, "bnneck.running_mean"), _t(sd, "bnneck.running_var"),
                           _t(sd, "bnneck.weight"), _t(sd, "bnneck.bias"), False, 0.0, BN_EPS)
        if cam is not None:
            emb = emb + cam_factor * _t(sd, "cam_bias")[torch.as_tensor(np.asarray(cam), dtype=torch.long)]
        logits = emb @ _t(sd, "classifier.0.weight").t()
    return emb, logits


def embed_u8(sd, crops_u8, bs=64):
    """Batch-64 embedding loop (reference default --bs 64, reid/image_reid_inference.py:144)."""
    outs = []
    for i in range(0, len(crops_u8), bs):
        outs.append(forward(sd, preprocess_u8(crops_u8[i:i + bs]))[0])
    return torch.cat(outs, 0).numpy()
