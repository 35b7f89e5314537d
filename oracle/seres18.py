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


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _bn(sd, prefix, x):
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


def _block(sd, name, ibn, ds, stride, x, taps=None):
    pre = name + ".block_pre"
    c1 = F.conv2d(x, _t(sd, pre + ".conv1.weight"), None, stride, 1)
    h = F.relu(_ibn(sd, pre + ".bn1", c1) if ibn else _bn(sd, pre + ".bn1", c1))
    y = _bn(sd, pre + ".bn2", F.conv2d(h, _t(sd, pre + ".conv2.weight"), None, 1, 1))
    if ds:
        sc = _bn(sd, name + ".block_post.bn", F.conv2d(x, _t(sd, name + ".block_post.conv.weight"), None, stride, 0))
    else:
        y = F.relu(y + x)          # BasicBlock_IBN's own residual + relu (block_pre is the whole block)
        sc = x
    s = _se(sd, name + ".seblock", y)
    out = F.relu(s * y + sc)
    if taps is not None:
        taps[name + ".conv1"] = c1
        taps[name + ".y"] = y
        taps[name + ".se"] = s.reshape(s.shape[0], -1)
        taps[name] = out
    return out


def preprocess_u8(crops_u8):
    """uint8[N,H,W,3] crops already at 128x256 -> float32[N,3,H,W], (x/255-0.5)/0.5.
    feature_extractor.py:40-46 (cv2.resize to the same size is the identity)."""
    x = torch.from_numpy(np.ascontiguousarray(crops_u8)).to(torch.float32) / 255.0
    x = (x - 0.5) / 0.5
    return x.permute(0, 3, 1, 2).contiguous()


def forward(sd, x, taps=None):
    """x: float32[N,3,256,128] NCHW (normalised).  Returns (emb[N,512], logits[N,num_class])."""
    with torch.no_grad():
        x = F.conv2d(x, _t(sd, "conv0.weight"), None, 2, 3)
        x = _bn(sd, "bn0", x)                                  # no ReLU (SERes18_IBN.py:253)
        if taps is not None:
            taps["stem"] = x
        x = F.max_pool2d(x, 3, 2, 1)
        if taps is not None:
            taps["pool0"] = x
        for name, _c, ibn, ds, stride in BLOCKS:
            x = _block(sd, name, ibn, ds, stride, x, taps)
        p = _t(sd, "avgpooling.p")
        feat = x.clamp(min=GEM_EPS).pow(p).mean(dim=(2, 3)).pow(1.0 / p)
        if taps is not None:
            taps["gem"] = feat
        emb = F.batch_norm(feat, _t(sd, "bnneck.running_mean"), _t(sd, "bnneck.running_var"),
                           _t(sd, "bnneck.weight"), _t(sd, "bnneck.bias"), False, 0.0, BN_EPS)
        logits = emb @ _t(sd, "classifier.0.weight").t()
    return emb, logits


def embed_u8(sd, crops_u8, bs=64):
    """Batch-64 embedding loop (reference default --bs 64, reid/image_reid_inference.py:144)."""
    outs = []
    for i in range(0, len(crops_u8), bs):
        outs.append(forward(sd, preprocess_u8(crops_u8[i:i + bs]))[0])
    return torch.cat(outs, 0).numpy()
