"""MI355X-native re-ID embedding-and-matching engine (DeepSORT hot path).

Host side mirrors the plugin surface of SuperbTUM/real-time-ReID-tracking:
  * ``Extractor``                      <- modification_deepsort/feature_extractor.py:14-53
  * ``reid_model_factory`` functions   <- modification_tracking/reid_model_factory.py
  * ``build_model`` registry           <- modification_tracking/models/__init__.py:93-121
  * ``euclidean_dist`` / ``cosine_dist``<- reid/losses/utils.py:12-35
  * ``evaluate_all``                   <- reid/evaluate.py:33-105
  * ``iou`` (DIoU)                     <- modification_deepsort/iou_matching.py:5-47
All arithmetic runs in hand-written HIP kernels behind the C ABI declared in
``include/reid_hip.h``; there is no CPU fallback (a missing ``libreid_hip.so``
raises at first use).
"""
__version__ = "0.1.0"
