"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the reference's custom Swin-T (version "v1") eval-mode forward as plain functional torch ops
on a ``state_dict`` - no einops, no timm, no ``nn.Module`` from the reference.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.  Pinned against the
reference's own class (run here with a stubbed ``timm``) by ``oracle/gen_golden.py`` -> ``tests/golden/swin_*.npz``.

Follows reid/backbones/swin_transformer.py (paths under /root/reference):
  :278-304  ShadowFeatureExtraction (conv2x2s2 -> MixedNorm(IN||BN) -> ReLU -> conv2x2s2 -> ReLU -> Linear 48->96)
  :263-275  PatchMerging (Unfold k2 s2: feature order (c, kh, kw) -> Linear)
  :117-232  WindowAttention v1 (roll(-3,-3), qkv, 7x7 windows, 32-d heads, scale 32^-0.5, relative position bias from a
            13x13 table shared by all heads, -inf masks on the last window row / column, to_out, post_proj, roll back)
  :235-260  SwinBlock v1: x + Attn(LN(x)); x + FF(LN(x)),  FF = Linear -> GELU(erf) -> Linear
  :397-427  forward: stages, top-down fusion (Conv 8x8 s8, 3x ConvTranspose 4x4 s2 p1), LN(eps 1e-6), GeM_1D, BN1d
Eval returns (logits, x_norm) in the reference; here (embedding[N,96], logits[N,num_class]).
"""
import numpy as np
import torch
import torch.nn.functional as F

WS, HEAD_DIM = 7, 32
DIMS, LAYERS, HEADS = (96, 192, 384, 768), (2, 2, 6, 2), (3, 6, 12, 24)


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _lin(sd, prefix, x, bias=True):
    return F.linear(x, _t(sd, prefix + ".weight"), _t(sd, prefix + ".bias") if bias else None)


def _attention(sd, prefix, x, heads, shifted):
    """x: [b, H, W, C] (already layer-normed)."""
    b, H, W, C = x.shape
    if shifted:
        x = torch.roll(x, shifts=(-3, -3), dims=(1, 2))
    qkv = _lin(sd, prefix + ".to_qkv", x, bias=False)
    nh, nw = H // WS, W // WS

    def split(t):   # 'b (nw_h w_h) (nw_w w_w) (h d) -> b h (nw_h nw_w) (w_h w_w) d'
        t = t.reshape(b, nh, WS, nw, WS, heads, HEAD_DIM)
        return t.permute(0, 5, 1, 3, 2, 4, 6).reshape(b, heads, nh * nw, WS * WS, HEAD_DIM)

    q, k, v = (split(t) for t in qkv.chunk(3, dim=-1))
    dots = torch.matmul(q, k.transpose(-1, -2)) * (HEAD_DIM ** -0.5)
    idx = torch.tensor([[y, xx] for y in range(WS) for xx in range(WS)])
    rel = idx[None, :, :] - idx[:, None, :] + WS - 1
    dots = dots + _t(sd, prefix + ".pos_embedding")[rel[:, :, 0], rel[:, :, 1]]
    if shifted:
        dots[:, :, -nw:] += _t(sd, prefix + ".upper_lower_mask")
        dots[:, :, nw - 1::nw] += _t(sd, prefix + ".left_right_mask")
    out = torch.matmul(dots.softmax(dim=-1), v)
    out = out.reshape(b, heads, nh, nw, WS, WS, HEAD_DIM).permute(0, 2, 4, 3, 5, 1, 6).reshape(b, H, W, C)
    out = _lin(sd, prefix + ".post_proj", _lin(sd, prefix + ".to_out", out))
    if shifted:
        out = torch.roll(out, shifts=(3, 3), dims=(1, 2))
    return out


def _block(sd, prefix, x, heads, shifted):
    c = x.shape[-1]
    a = prefix + ".attention_block.fn"
    h = F.layer_norm(x, (c,), _t(sd, a + ".norm.weight"), _t(sd, a + ".norm.bias"), 1e-5)
    x = x + _attention(sd, a + ".fn", h, heads, shifted)
    m = prefix + ".mlp_block.fn"
    h = F.layer_norm(x, (c,), _t(sd, m + ".norm.weight"), _t(sd, m + ".norm.bias"), 1e-5)
    h = _lin(sd, m + ".fn.net.3", F.gelu(_lin(sd, m + ".fn.net.0", h)))
    return x + h


def forward(sd, img, taps=None, view_index=None, side_info_coeff=1.5):
    """img: float32[N,3,224,224] (or any size whose /4 grid is divisible by 7 down to /32).  -> (emb[N,96], logits).
    ``view_index`` (per image) adds side_info_coeff * side_info_embedding[view] to the SFE output (swin_transformer.py:298-302;
    the coefficient is the constructor's, default 1.5, :279)."""
    with torch.no_grad():
        x = F.conv2d(img, _t(sd, "sfe.conv1.weight"), _t(sd, "sfe.conv1.bias"), 2)
        a = F.instance_norm(x[:, :6].contiguous(), None, None, _t(sd, "sfe.norm.instancenorm.weight"),
                            _t(sd, "sfe.norm.instancenorm.bias"), True, 0.0, 1e-5)
        bb = F.batch_norm(x[:, 6:].contiguous(), _t(sd, "sfe.norm.batchnorm.running_mean"),
                          _t(sd, "sfe.norm.batchnorm.running_var"), _t(sd, "sfe.norm.batchnorm.weight"),
                          _t(sd, "sfe.norm.batchnorm.bias"), False, 0.0, 1e-5)
        x = F.relu(torch.cat((a, bb), 1))
        x = F.relu(F.conv2d(x, _t(sd, "sfe.conv2.weight"), _t(sd, "sfe.conv2.bias"), 2))
        sfe = _lin(sd, "sfe.fc", x.permute(0, 2, 3, 1))                     # [N,56,56,96] NHWC
        if view_index is not None:
            sfe = sfe + side_info_coeff * _t(sd, "sfe.side_info_embedding")[torch.as_tensor(np.asarray(view_index), dtype=torch.long)]
        if taps is not None:
            taps["sfe"] = sfe
        outs = []
        x = sfe
        for si in range(4):
            st = "stage%d" % (si + 1)
            if si > 0:   # PatchMerging: Unfold(k=2, s=2) feature order is (c, kh, kw)
                n, h, w, c = x.shape
                u = x.reshape(n, h // 2, 2, w // 2, 2, c).permute(0, 1, 3, 5, 2, 4).reshape(n, h // 2, w // 2, c * 4)
                x = _lin(sd, st + ".patch_partition.linear", u)
            for li in range(LAYERS[si] // 2):
                x = _block(sd, "%s.layers.%d.0" % (st, li), x, HEADS[si], False)
                x = _block(sd, "%s.layers.%d.1" % (st, li), x, HEADS[si], True)
            outs.append(x)
            if taps is not None:
                taps[st] = x
        nchw = [o.permute(0, 3, 1, 2) for o in outs]
        f = nchw[3] + F.conv2d(sfe.permute(0, 3, 1, 2), _t(sd, "img_channel_align.weight"), _t(sd, "img_channel_align.bias"), 8)
        f = nchw[2] + F.conv_transpose2d(f, _t(sd, "stage4_channel_align.weight"), _t(sd, "stage4_channel_align.bias"), 2, 1)
        f = F.conv_transpose2d(f, _t(sd, "stage3_channel_align.weight"), _t(sd, "stage3_channel_align.bias"), 2, 1) + nchw[1]
        f = F.conv_transpose2d(f, _t(sd, "stage2_channel_align.weight"), _t(sd, "stage2_channel_align.bias"), 2, 1) + nchw[0]
        if taps is not None:
            taps["fused"] = f.permute(0, 2, 3, 1)
        tok = f.flatten(2).permute(0, 2, 1)                                 # [N, L, 96]
        tok = F.layer_norm(tok, (96,), _t(sd, "norm.weight"), _t(sd, "norm.bias"), 1e-6)
        p = _t(sd, "avgpool.p")
        g = tok.clamp(min=1e-6).pow(p).mean(dim=1).pow(1.0 / p)             # GeM_1D over the tokens
        if taps is not None:
            taps["gem"] = g
        emb = F.batch_norm(g, _t(sd, "bottleneck.running_mean"), _t(sd, "bottleneck.running_var"),
                           _t(sd, "bottleneck.weight"), _t(sd, "bottleneck.bias"), False, 0.0, 1e-5)
        logits = F.linear(emb, _t(sd, "mlp_head.0.weight"))
    return emb, logits


def embed(sd, imgs, bs=16):
    outs = []
    for i in range(0, len(imgs), bs):
        outs.append(forward(sd, torch.from_numpy(np.ascontiguousarray(imgs[i:i + bs])))[0])
    return torch.cat(outs, 0).numpy()
