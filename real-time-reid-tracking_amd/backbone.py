"""Backbone objects handed out by ``build_model`` - what the tracker calls as ``model(im_batch)``.

They hold the checkpoint as a plain numpy ``state_dict`` (same keys as the reference's module,
reid/backbones/SERes18_IBN.py:186-248) and a handle to the HIP engine; there is no torch graph and no CPU
forward.  They tolerate the calls the external loader makes on an ``nn.Module``: ``.to(device)``, ``.half()``,
``.eval()``, ``state_dict()`` / ``load_state_dict()``, ``warmup()``.
"""
from collections import OrderedDict

import numpy as np

from . import precision as _precision
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


def _copy_matching(dst, src, strict, resizable=()):
    """Copies checkpoint tensors into the held state_dict.  A shape mismatch raises when ``strict`` (as torch does:
    "size mismatch for ..."); otherwise the tensor is skipped with a warning naming it - except the classifier, whose row
    count (num_classes of the training set) is taken from the checkpoint.  Returns the skipped keys."""
    bad = []
    for k, v in src.items():
        if k not in dst:
            continue
        v = np.asarray(v)
        if tuple(dst[k].shape) == tuple(v.shape):
            dst[k] = v.astype(dst[k].dtype)
        elif k in resizable and v.ndim == dst[k].ndim and v.shape[1:] == dst[k].shape[1:]:
            dst[k] = v.astype(dst[k].dtype)             # a checkpoint trained with another number of identities
        else:
            bad.append("%s: checkpoint %s vs model %s" % (k, tuple(v.shape), tuple(dst[k].shape)))
    if bad:
        if strict:
            raise RuntimeError("Error(s) in loading state_dict: size mismatch for " + "; ".join(bad[:5]))
        import warnings
        warnings.warn("load_state_dict(strict=False): skipped tensors with another shape (seeded values kept): " + "; ".join(bad))
    return [b.split(":")[0] for b in bad]


def _torch_cuda_forward(x, embed_dev, engine, out_dims):
    """CUDA tensor in -> CUDA tensors out, no host round trip: the device entry point runs on ``x.data_ptr()`` on torch's
    current stream of that device (the yolov8_tracking multibackend hands the model CUDA batches every frame)."""
    import torch
    xf = x.detach()
    if xf.dtype != torch.float32 or not xf.is_contiguous():
        xf = xf.float().contiguous()
    with torch.cuda.device(xf.device), engine.on_torch_stream():
        outs = [torch.empty((xf.shape[0], d), dtype=torch.float32, device=xf.device) for d in out_dims]
        embed_dev(xf, outs)
    return outs


def _index_array(index, n):
    """cam / view_index argument (sequence, numpy array or torch tensor, one entry per image) -> int32[n]."""
    a = index.detach().cpu().numpy() if hasattr(index, "detach") else np.asarray(index)
    a = a.reshape(-1).astype(np.int64)
    if a.shape[0] != n:
        raise ValueError("expected one index per image (%d), got %d" % (n, a.shape[0]))
    return a.astype(np.int32)


class SERes18IBN:
    """ResNet18-IBN-a + SE + GeM + BNNeck ("ResNet18-SE"), eval mode only.

    ``model(x)`` -> embedding [N,512] = BNNeck output; ``model(x, return_logits=True)`` ->
    (embedding, logits[N,num_classes]) like the reference's eval-mode tuple (SERes18_IBN.py:274-275).
    ``x``: float32/float16 [N,3,256,128] NCHW, normalised - numpy array or torch tensor; the result has the
    type (and, for torch, the device) of the input.
    """

    embed_dim = 512
    arch = "seres18_ibn"

    def __init__(self, num_classes=751, loss="triplet", pretrained=False, use_gpu=True, seed=0, num_cams=6, cam_factor=-1.0,
                 precision=None, **_):
        if loss not in ("triplet", "softmax"):
            raise NotImplementedError                      # seres18_ibn(), SERes18_IBN.py:279-285
        self._mode = _precision.resolve(precision)         # "f16x3" unless the argument or $REID_PRECISION says otherwise (precision.py)
        self.num_classes = num_classes
        self.num_cams, self.cam_factor = num_cams, float(cam_factor)   # SERes18_IBN.py:193,198: cam_bias [num_cams,512] and its factor
        self.is_reid = loss == "softmax"
        self.training = False
        self._device = 0
        # held with SERse18_IBN's named keys; state_dict() hands out the reference's own key layout (positional for the siblings'
        # downsample blocks), load_state_dict() accepts both
        self._sd = weights.normalize_state_dict(synth.seres18_state_dict(seed, num_class=num_classes, num_cams=num_cams, gem_p=3.0,
                                                                         arch=self.arch))
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

    def _ref_key(self, k):
        ds = ("basicBlock21", "basicBlock31", "basicBlock41")
        return synth.sibling_key(k) if getattr(self, "arch", "seres18_ibn") != "seres18_ibn" and k.startswith(ds) else k

    def state_dict(self):
        try:
            import torch
            return OrderedDict((self._ref_key(k), torch.from_numpy(np.array(v))) for k, v in self._sd.items())
        except ImportError:
            return OrderedDict((self._ref_key(k), np.array(v)) for k, v in self._sd.items())

    def load_state_dict(self, state_dict, strict=True):
        sd = weights.normalize_state_dict(state_dict)
        missing = [k for k in self._sd if k not in sd]
        unexpected = [k for k in sd if k not in self._sd]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict: missing %s unexpected %s" % (missing[:5], unexpected[:5]))
        mismatched = _copy_matching(self._sd, sd, strict, resizable=("classifier.0.weight",))
        if "classifier.0.weight" in self._sd:
            self.num_classes = self._sd["classifier.0.weight"].shape[0]
        self._dirty = True
        return missing, unexpected + mismatched

    # ---- engine
    @property
    def precision(self):
        """The arithmetic this model runs in now ("f16x3", "f32" or "f16"; "f32" after a fall back, precision.py)."""
        return _precision.LABEL[self._mode]

    def _needs_bind(self, eng):
        return self._dirty or getattr(eng, "_owner", None) is not self

    def _do_bind(self, eng):
        blob, manifest, _ = weights.pack_seres18(self._sd, cam_factor=getattr(self, "cam_factor", -1.0))
        eng.load_seres18(blob, manifest)
        eng._owner = self
        self._dirty = False

    def _run(self, fn):
        """fn(engine) with this model's weights bound and the context in this model's arithmetic (precision.run)."""
        return _precision.run(self, get_engine(self._device), self.arch, fn)

    def warmup(self, imgsz=(1, 3, IMG_H, IMG_W)):
        """One dummy batch: allocates workspaces and loads code objects (track_yolov5.py:169-171)."""
        self(np.zeros(imgsz, np.float32))
        return self

    def __call__(self, x, cam=None, return_logits=False):
        """``cam``: camera index per image (int array / tensor) - adds cam_factor * cam_bias[cam] to the embedding before the
        classifier, as SERse18_IBN.forward(x, cam) does (SERes18_IBN.py:269-271); the extractor path never passes it."""
        try:
            return self._forward(x, cam, return_logits)
        finally:
            if cam is not None:
                get_engine(self._device).set_side_index(None)     # nothing pending after a failed call

    def _forward(self, x, cam, return_logits):
        is_torch = hasattr(x, "detach")
        if not is_torch:
            x = np.asarray(x, np.float32)
        if x.ndim != 4 or tuple(x.shape[1:]) != (3, IMG_H, IMG_W):
            raise ValueError("expected [N,3,%d,%d] input, got %s" % (IMG_H, IMG_W, tuple(x.shape)))
        if is_torch and x.is_cuda:
            if x.device.index != self._device:
                self.to(x.device.index)


            def on_device(eng):
                if cam is not None:
                    eng.set_side_index(_index_array(cam, x.shape[0]))
                return _torch_cuda_forward(
                    x, lambda xf, o: eng.embed_f32_nchw_dev(xf.data_ptr(), xf.shape[0], o[0].data_ptr(), o[1].data_ptr()), eng,
                    (self.embed_dim, self.num_classes))
            emb, logits = self._run(on_device)
        else:
            x_np = x.detach().float().cpu().numpy() if is_torch else np.asarray(x, np.float32)

            def on_host(eng):
                if cam is not None:
                    eng.set_side_index(_index_array(cam, x_np.shape[0]))
                return eng.embed_f32_nchw(x_np, logits=True)
            emb, logits = self._run(on_host)
            if is_torch:
                import torch
                emb, logits = torch.from_numpy(emb), torch.from_numpy(logits)
        if self.is_reid:
            return logits                                   # SERes18_IBN.py:272-273
        return (emb, logits) if return_logits else emb

    forward = __call__

    def embed_crops(self, crops):
        """uint8 HxWx3 crops of any size -> float32[N,512]; resize + normalise on the device."""
        crops = list(crops)
        return self._run(lambda eng: eng.embed_ragged_u8(crops))


def seres18_ibn(num_classes=751, loss="triplet", pretrained=False, use_gpu=True, **kwargs):
    """Factory with the registry's calling convention (models/__init__.py:116-121)."""
    return SERes18IBN(num_classes=num_classes, loss=loss, pretrained=pretrained, use_gpu=use_gpu, **kwargs)


class CARes18IBN(SERes18IBN):
    """CARes18_IBN (reid/backbones/CARes18.py:185-281): the same IBN-Net skeleton, every block ends in
    relu(TripletAttention(y) + shortcut) (:148-157; triplet_attention.py:69-101).  Exact fp32 arithmetic only."""
    arch = "cares18_ibn"


class EMARes18IBN(SERes18IBN):
    """EMARes18_IBN (reid/backbones/EMA_Res18.py:118-181): every block ends in relu(EMA(y) + shortcut) (:79-86), EMA with 32
    channel groups (:10-38).  Exact fp32 arithmetic only."""
    arch = "emares18_ibn"


def cares18_ibn(num_classes=751, loss="triplet", pretrained=False, use_gpu=True, **kwargs):
    """Same name as the reference's factory (CARes18.py:270-281)."""
    return CARes18IBN(num_classes=num_classes, loss=loss, pretrained=pretrained, use_gpu=use_gpu, **kwargs)


def emares18_ibn(num_classes=751, loss="triplet", pretrained=False, use_gpu=True, **kwargs):
    """Same name as the reference's factory (EMA_Res18.py:184-195)."""
    return EMARes18IBN(num_classes=num_classes, loss=loss, pretrained=pretrained, use_gpu=use_gpu, **kwargs)


class SwinT:
    """The reference's custom Swin-T, version "v1" (reid/backbones/swin_transformer.py:339-427, swin_t :508-513): conv stem,
    4 stages of (W-MSA, SW-MSA) blocks, top-down ConvTranspose fusion, LN -> GeM_1D -> BatchNorm1d(96).

    ``model(x)`` -> embedding [N,96] (x_norm); ``model(x, return_logits=True)`` -> (logits, embedding), the order of the
    reference's eval-mode tuple (:422-423).  ``x``: float [N,3,H,W] with H, W multiples of 224 - the
    reference itself rejects 128x256.  Unlike the reference (Q14) N = 1 works.
    """

    embed_dim = 96

    arch = "swin_transformer"
    _arch = "swin"            # precision.run: which of the shared engine's two weight sets this object owns

    def __init__(self, num_classes=751, loss="softmax", pretrained=False, use_gpu=True, seed=0, camera=0, sequence=0, side_info=True,
                 side_info_coeff=1.5, precision=None, **_):
        self._mode = _precision.resolve(precision)
        self.num_classes = num_classes
        self.loss = loss
        self.training = False
        self._device = 0
        # ShadowFeatureExtraction's side-information table (swin_transformer.py:285-293): camera * sequence, camera or sequence rows
        self.views = camera * sequence if camera * sequence > 0 else camera if camera > 0 else max(sequence, 0)
        self.side_info, self.side_info_coeff = bool(side_info), float(side_info_coeff)
        self._sd = synth.swin_state_dict(seed, num_class=num_classes, views=self.views)
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
    _ref_key = SERes18IBN._ref_key

    def load_state_dict(self, state_dict, strict=True):
        sd = weights.normalize_state_dict(state_dict)
        missing = [k for k in self._sd if k not in sd]
        unexpected = [k for k in sd if k not in self._sd]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict: missing %s unexpected %s" % (missing[:5], unexpected[:5]))
        mismatched = _copy_matching(self._sd, sd, strict, resizable=("classifier.weight", "classifier.0.weight"))
        for k in ("classifier.weight", "classifier.0.weight"):
            if k in self._sd:
                self.num_classes = self._sd[k].shape[0]
        self._dirty = True
        return missing, unexpected + mismatched

    precision = SERes18IBN.precision
    _run = SERes18IBN._run

    def _needs_bind(self, eng):
        return self._dirty or getattr(eng, "_swin_owner", None) is not self

    def _do_bind(self, eng):
        blob, manifest, _ = weights.pack_swin(self._sd, side_info_coeff=getattr(self, "side_info_coeff", 1.5))
        eng.load_swin(blob, manifest)
        eng._swin_owner = self
        self._dirty = False

    def warmup(self, imgsz=(1, 3, 224, 224)):
        self(np.zeros(imgsz, np.float32))
        return self

    def __call__(self, x, view_index=None, return_logits=False):
        """``view_index``: view (camera / sequence) index per image - adds side_info_coeff * side_info_embedding[view] to the SFE
        output when the model was built with side_info and camera / sequence > 0 (swin_transformer.py:301-302, which ignores
        the argument otherwise - so does this)."""
        if view_index is not None and not (self.side_info and self.views > 0):
            if self.side_info:      # the reference would index a parameter that does not exist (AttributeError)
                raise AttributeError("SwinTransformer was built without camera / sequence: no side_info_embedding")
            view_index = None
        try:
            return self._forward(x, view_index, return_logits)
        finally:
            if view_index is not None:
                get_engine(self._device).set_side_index(None)

    def _forward(self, x, view_index, return_logits):
        is_torch = hasattr(x, "detach")
        if is_torch and x.is_cuda:
            if x.ndim != 4 or x.shape[1] != 3 or x.shape[2] % 224 or x.shape[3] % 224:
                raise ValueError("expected float[n,3,224k,224m], got %s" % (tuple(x.shape),))
            if x.device.index != self._device:
                self.to(x.device.index)


            def on_device(eng):
                if view_index is not None:
                    eng.set_side_index(_index_array(view_index, x.shape[0]))
                return _torch_cuda_forward(
                    x, lambda xf, o: eng.swin_embed_dev(xf.data_ptr(), xf.shape[0], xf.shape[2], xf.shape[3], o[0].data_ptr(), o[1].data_ptr()),
                    eng, (self.embed_dim, self.num_classes))
            emb, logits = self._run(on_device)
        else:
            x_np = x.detach().float().cpu().numpy() if is_torch else np.asarray(x, np.float32)

            def on_host(eng):
                if view_index is not None:
                    eng.set_side_index(_index_array(view_index, x_np.shape[0]))
                return eng.swin_embed_f32_nchw(x_np, logits=True)
            emb, logits = self._run(on_host)
            if is_torch:
                import torch
                emb, logits = torch.from_numpy(emb), torch.from_numpy(logits)
        return (logits, emb) if return_logits else emb

    forward = __call__


def swin_t(num_classes=751, loss="softmax", pretrained=False, use_gpu=True, **kwargs):
    """Registry constructor, same name as the reference's (swin_transformer.py:508)."""
    return SwinT(num_classes=num_classes, loss=loss, pretrained=pretrained, use_gpu=use_gpu, **kwargs)
