"""DeepSORT appearance extractor - mirror of modification_deepsort/feature_extractor.py:14-53.

    extractor = Extractor(model_path, use_cuda=True)
    features  = extractor(im_crops)        # list of HxWx3 uint8 -> np.ndarray float32 [N, 512]

What the reference does per call (feature_extractor.py:31-53): a Python loop of cv2.resize + ToTensor + Normalize
per crop, torch.cat, H2D copy, backbone forward, D2H copy.  Here the crops are packed once, copied once, and the
resize / normalise / backbone all run as HIP kernels; the return value is a fresh C-contiguous host array.

Deviations, all forced by defects of the reference file itself:
  Q1  it imports a class that no longer exists (SEDense18_IBN); this binds to the current SERse18_IBN layout.
  Q2  it never calls .eval(); the only well-defined semantics, used here, is eval mode (running-stat BN).
  Q3  embedding = BNNeck output (512-d), the first element of the eval-mode tuple (SERes18_IBN.py:274-275).
"""
import logging

import numpy as np

from . import precision as _precision
from . import weights


class Extractor(object):
    def __init__(self, model_path, use_cuda=True, device=0, precision=None):
        """``Extractor(model_path, use_cuda=True)`` as feature_extractor.py:15-29.  ``precision`` (keyword, not in the reference):
        "f16x3" (default; $REID_PRECISION overrides the default) / "f32" / "f16" - see precision.py; a checkpoint the fp32-class
        arithmetic cannot represent runs in exact fp32, with one log line."""
        if not use_cuda:
            raise RuntimeError("Extractor: the MI355X engine has no CPU path (use_cuda=False is not supported)")
        self.device = "cuda"
        self.size = (128, 256)                 # (W, H), feature_extractor.py:24
        self._mode = _precision.resolve(precision)
        if isinstance(model_path, dict):
            state_dict = model_path            # already-loaded state_dict (tests, benchmarks)
        else:
            # PyTorch only reads the checkpoint (feature_extractor.py:18) - and it is imported BEFORE the engine: torch bundles its
            # own HIP runtime, and the process must hold ONE (libreid_hip.so binds to whichever is loaded first; the other order
            # loads two and aborts in the exit handlers - DESIGN.md section 6, the rule parallel.RcclComm.from_env enforces)
            import torch
            state_dict = torch.load(model_path, map_location="cpu")   # {"state_dict": ...} / "module." prefixes: weights.pack_seres18
        from .engine import get_engine
        self.net = get_engine(device)
        blob, manifest, self.info = weights.pack_seres18(state_dict)   # strict=False semantics, :19
        self._packed = (blob, manifest)
        self.net._owner = None
        _precision.run(self, self.net, "Extractor", lambda eng: None)      # load now; a checkpoint mode 2 refuses falls back here
        logger = logging.getLogger("root.tracker")
        logger.info("Loading weights from {}... Done!".format(model_path if not isinstance(model_path, dict) else "<state_dict>"))

    @property
    def precision(self):
        """The arithmetic this extractor runs in now ("f16x3", "f32" or "f16"; "f32" after a fall back)."""
        return _precision.LABEL[self._mode]

    def _preprocess(self, im_crops):
        """Host-visible equivalent of feature_extractor.py:31-46 is fused into the device path; this only
        validates the crops the way the reference's torch.cat would fail on an empty list."""
        if len(im_crops) == 0:
            raise RuntimeError("Extractor: expected a non-empty list of crops")   # torch.cat([]) raises, :44
        return [np.asarray(im) for im in im_crops]

    def _needs_bind(self, eng):
        return getattr(eng, "_owner", None) is not self        # another model was loaded on this device meanwhile

    def _do_bind(self, eng):
        eng.load_seres18(*self._packed)
        eng._owner = self

    def _run(self, fn):
        return _precision.run(self, self.net, "Extractor", fn)

    def __call__(self, im_crops):
        """feature_extractor.py:48-53: host crops in, host features out (the upload of one pass of crops, its download and the
        kernels of its neighbours overlap inside the library).  A stacked ``uint8[n,256,128,3]`` array - crops already at the
        extractor's size, for which the reference's cv2.resize is the identity - goes to the fixed-size entry point as it is."""
        if isinstance(im_crops, np.ndarray) and im_crops.dtype == np.uint8 and im_crops.ndim == 4 \
                and im_crops.shape[1:] == (self.size[1], self.size[0], 3) and im_crops.shape[0] > 0:
            return self._run(lambda eng: eng.embed_u8(im_crops))
        crops = self._preprocess(im_crops)
        return self._run(lambda eng: eng.embed_ragged_u8(crops))

    def from_frame(self, bbox_xywh, ori_img):
        """DeepSort._get_features(bbox_xywh, ori_img) ([external] deep_sort.py) in one call: centre-format boxes are
        converted and clipped like DeepSort._xywh_to_xyxy (x1 = max(int(x - w/2), 0), x2 = min(int(x + w/2), W - 1), ...),
        the frame goes to the device once and the windows are cut and resized there.  Returns float32 [N, 512]
        (an empty array when there is no box, as the reference's `np.array([])` branch)."""
        ori_img = np.asarray(ori_img)
        boxes = np.asarray(bbox_xywh, dtype=np.float64).reshape(-1, 4)
        if boxes.shape[0] == 0:
            return np.array([])
        height, width = ori_img.shape[:2]
        xyxy = np.empty((boxes.shape[0], 4), np.int32)
        for i, (x, y, w, h) in enumerate(boxes):
            xyxy[i] = (max(int(x - w / 2), 0), max(int(y - h / 2), 0), min(int(x + w / 2), width - 1), min(int(y + h / 2), height - 1))
        return self._run(lambda eng: eng.embed_frame_u8(ori_img, xyxy))
