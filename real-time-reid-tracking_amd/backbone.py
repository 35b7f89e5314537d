"""Backbone objects handed out by ``build_model`` - what the tracker calls as ``model(im_batch)``.

They hold the checkpoint as a plain numpy ``state_dict`` (same keys as the reference's module,
reid/backbones/SERes18_IBN.py:186-248) and a handle to the HIP engine; there is no torch graph and no CPU
forward.  They tolerate the calls the external loader makes on an ``nn.Module``: ``.to(device)``, ``.half()``,
``.eval()``, ``state_dict()`` / ``load_state_dict()``, ``warmup()``.
"""
from collections import OrderedDict

import numpy as np

from . import synth, weights
from .engine import IMG_H, IMG_W, get_engine


def _device_index(device):
    if device is None:
        return 0
    if isinstance(device, int):
        return device
    s = str(device)
    if s in ("cuda", "gpu", "hip"):
        return 0
    if s.startswith("cuda:"):
        return int(s.split(":")[1])
    if s == "cpu":
        raise RuntimeError("this backbone runs only on an MI355X through libreid_hip.so; there is no CPU path")
    return int(s)


class SERes18IBN:
    """ResNet18-IBN-a + SE + GeM + BNNeck ("ResNet18-SE"), eval mode only.

    ``model(x)`` -> embedding [N,512] = BNNeck output; ``model(x, return_logits=True)`` ->
    (embedding, logits[N,num_classes]) like the reference's eval-mode tuple (SERes18_IBN.py:274-275).
    ``x``: float32/float16 [N,3,256,128] NCHW, normalised - numpy array or torch tensor; the result has the
    type (and, for torch, the device) of the input.
    """

    embed_dim = 512

    def __init__(self, num_classes=751, loss="triplet", pretrained=False, use_gpu=True, seed=0, **_):
        if loss not in ("triplet", "softmax"):
            raise NotImplementedError                      # seres18_ibn(), SERes18_IBN.py:279-285
        self.num_classes = num_classes
        self.is_reid = loss == "softmax"
        self.training = False
        self._device = 0
        self._sd = synth.seres18_state_dict(seed, num_class=num_classes, gem_p=3.0)
        self._dirty = True
        if pretrained:
            import warnings
            warnings.warn("ImageNet-pretrained IBN-Net weights need torch.hub/network access (SERes18_IBN.py:201); "
                          "keeping the seeded random initialisation")

    # ---- nn.Module-like surface
    def to(self, device=None, *a, **k):
        idx = _device_index(device)
        if idx != self._device:
            self._device, self._dirty = idx, True
        return self

    def cuda(self, device=None):
        return self.to(0 if device is None else device)

    def half(self):
        return self            # fp16 inputs are widened on entry; arithmetic stays fp32

    def float(self):
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("inference engine: eval mode only")
        return self

    def parameters(self):
        return iter(())

    def state_dict(self):
        try:
            import torch
            return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in self._sd.items())
        except ImportError:
            return OrderedDict((k, np.array(v)) for k, v in self._sd.items())

    def load_state_dict(self, state_dict, strict=True):
        sd = weights.normalize_state_dict(state_dict)
        missing = [k for k in self._sd if k not in sd]
        unexpected = [k for k in sd if k not in self._sd]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict: missing %s unexpected %s" % (missing[:5], unexpected[:5]))
        for k, v in sd.items():
            if k in self._sd and tuple(self._sd[k].shape) == tuple(v.shape):
                self._sd[k] = np.asarray(v).astype(self._sd[k].dtype)
        if "classifier.0.weight" in self._sd:
            self.num_classes = self._sd["classifier.0.weight"].shape[0]
        self._dirty = True
        return missing, unexpected

    # ---- engine
    def _engine(self):
        eng = get_engine(self._device)
        if self._dirty or getattr(eng, "_owner", None) is not self:
            blob, manifest, _ = weights.pack_seres18(self._sd)
            eng.load_seres18(blob, manifest)
            eng._owner = self
            self._dirty = False
        return eng

    def warmup(self, imgsz=(1, 3, IMG_H, IMG_W)):
        """One dummy batch: allocates workspaces and loads code objects (track_yolov5.py:169-171)."""
        self(np.zeros(imgsz, np.float32))
        return self

    def __call__(self, x, cam=None, return_logits=False):
        if cam is not None:
            raise NotImplementedError("camera-bias term (SERes18_IBN.py:269-270) is never used by the extractor path")
        is_torch = hasattr(x, "detach")
        if is_torch:
            dev = x.device
            x_np = x.detach().float().cpu().numpy()
        else:
            x_np = np.asarray(x, np.float32)
        if x_np.ndim != 4 or x_np.shape[1:] != (3, IMG_H, IMG_W):
            raise ValueError("expected [N,3,%d,%d] input, got %s" % (IMG_H, IMG_W, tuple(x_np.shape)))
        emb, logits = self._engine().embed_f32_nchw(x_np, logits=True)
        if is_torch:
            import torch
            emb, logits = torch.from_numpy(emb).to(dev), torch.from_numpy(logits).to(dev)
        if self.is_reid:
            return logits                                   # SERes18_IBN.py:272-273
        return (emb, logits) if return_logits else emb

    forward = __call__

    def embed_crops(self, crops):
        """uint8 HxWx3 crops of any size -> float32[N,512]; resize + normalise on the device."""
        return self._engine().embed_ragged_u8(list(crops))


def seres18_ibn(num_classes=751, loss="triplet", pretrained=False, use_gpu=True, **kwargs):
    """Factory with the registry's calling convention (models/__init__.py:116-121)."""
    return SERes18IBN(num_classes=num_classes, loss=loss, pretrained=pretrained, use_gpu=use_gpu, **kwargs)


class SwinT:
    """The reference's custom Swin-T, version "v1" (reid/backbones/swin_transformer.py:339-427, swin_t :508-513): conv stem,
    4 stages of (W-MSA, SW-MSA) blocks, top-down ConvTranspose fusion, LN -> GeM_1D -> BatchNorm1d(96).

    ``model(x)`` -> embedding [N,96] (x_norm); ``model(x, return_logits=True)`` -> (logits, embedding), the order of the
    reference's eval-mode tuple (:422-423).  ``x``: float [N,3,H,W] with H, W multiples of 224 - the
    reference itself rejects 128x256.  Unlike the reference (Q14) N = 1 works.
    """

    embed_dim = 96

    def __init__(self, num_classes=751, loss="softmax", pretrained=False, use_gpu=True, seed=0, **_):
        self.num_classes = num_classes
        self.loss = loss
        self.training = False
        self._device = 0
        self._sd = synth.swin_state_dict(seed, num_class=num_classes)
        self._dirty = True
        if pretrained:
            import warnings
            warnings.warn("no pretrained Swin weights are reachable offline (pretrained_urls is empty in the reference, "
                          "swin_transformer.py:19); keeping the seeded random initialisation")

    to = SERes18IBN.to
    cuda = SERes18IBN.cuda
    half = SERes18IBN.half
    float = SERes18IBN.float
    eval = SERes18IBN.eval
    train = SERes18IBN.train
    parameters = SERes18IBN.parameters
    state_dict = SERes18IBN.state_dict

    def load_state_dict(self, state_dict, strict=True):
        sd = weights.normalize_state_dict(state_dict)
        missing = [k for k in self._sd if k not in sd]
        unexpected = [k for k in sd if k not in self._sd]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict: missing %s unexpected %s" % (missing[:5], unexpected[:5]))
        for k, v in sd.items():
            if k in self._sd and tuple(self._sd[k].shape) == tuple(v.shape):
                self._sd[k] = np.asarray(v).astype(self._sd[k].dtype)
        self._dirty = True
        return missing, unexpected

    def _engine(self):
        eng = get_engine(self._device)
        if self._dirty or getattr(eng, "_swin_owner", None) is not self:
            blob, manifest, _ = weights.pack_swin(self._sd)
            eng.load_swin(blob, manifest)
            eng._swin_owner = self
            self._dirty = False
        return eng

    def warmup(self, imgsz=(1, 3, 224, 224)):
        self(np.zeros(imgsz, np.float32))
        return self

    def __call__(self, x, view_index=None, return_logits=False):
        if view_index is not None:
            raise NotImplementedError("side-information embedding (swin_transformer.py:301-302) is not used by the plugin path")
        is_torch = hasattr(x, "detach")
        if is_torch:
            dev = x.device
            x_np = x.detach().float().cpu().numpy()
        else:
            x_np = np.asarray(x, np.float32)
        emb, logits = self._engine().swin_embed_f32_nchw(x_np, logits=True)
        if is_torch:
            import torch
            emb, logits = torch.from_numpy(emb).to(dev), torch.from_numpy(logits).to(dev)
        return (logits, emb) if return_logits else emb

    forward = __call__


def swin_t(num_classes=751, loss="softmax", pretrained=False, use_gpu=True, **kwargs):
    """Registry constructor, same name as the reference's (swin_transformer.py:508)."""
    return SwinT(num_classes=num_classes, loss=loss, pretrained=pretrained, use_gpu=use_gpu, **kwargs)
