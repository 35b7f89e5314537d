"""Host mirror of DeepSORT's `nn_matching.NearestNeighborDistanceMetric` ([external] deep_sort/sort/nn_matching.py - the
object `DeepSort.__init__` builds from `MAX_DIST`/`NN_BUDGET`, modification_deepsort/deep_sort.yaml:3,9) with the feature
bank and the T x M cost matrix on the device (csrc/bank.hip).  Same constructor, `partial_fit` and `distance`; `samples`
is not a dict of Python lists any more but a ring buffer in HBM (`samples_count(target)` tells how many are held).
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import check
from .engine import get_engine


class NearestNeighborDistanceMetric:
    def __init__(self, metric, matching_threshold, budget=None, max_tracks=4096, device=0, engine=None):
        if metric == "euclidean":
            self._metric = _ffi.METRIC_L2SQR
        elif metric == "cosine":
            self._metric = _ffi.METRIC_COS
        else:
            raise ValueError("Invalid metric; must be either 'euclidean' or 'cosine'")
        self.matching_threshold = matching_threshold
        self.budget = budget
        # the reference keeps every sample when budget is None; the device ring needs a bound (oldest fall out past it)
        self._ring = int(budget) if budget is not None else 1024
        self._max_tracks = int(max_tracks)
        self._eng = engine if engine is not None else get_engine(device)   # engine: a camera stream's own context
        self._bank = None
        self._slot = {}            # target id -> slot
        self._free = list(range(self._max_tracks - 1, -1, -1))

    # ------------------------------------------------------------------ bank plumbing
    def _ensure(self, d):
        if self._bank is None:
            h = C.c_void_p()
            check(self._eng.lib.reid_bank_create(self._eng.h, self._max_tracks, self._ring, int(d), C.byref(h)))
            self._bank, self._d = h, int(d)
            if hasattr(self._eng, "register_bank"):
                self._eng.register_bank(self)
        elif d != self._d:
            raise ValueError(f"feature dimension changed from {self._d} to {d}")

    def close(self):
        """Free the device bank (before its engine is closed: a bank belongs to a context)."""
        if self._bank is not None and self._eng.h:
            self._eng.lib.reid_bank_destroy(self._bank)
        self._bank = None
        if hasattr(self._eng, "unregister_bank"):
            self._eng.unregister_bank(self)

    def __del__(self):
        # engine and bank reference each other: the collector may finalise either first, and the engine's close() closes its
        # banks - so this must leave the bank marked as gone (close() does), or it is destroyed twice
        try:
            self.close()
        except Exception:
            pass

    def samples_count(self, target):
        n = C.c_int()
        check(self._eng.lib.reid_bank_count(self._bank, self._slot[target], C.byref(n)))
        return n.value

    @property
    def targets(self):
        return list(self._slot)

    # ------------------------------------------------------------------ reference interface
    def partial_fit(self, features, targets, active_targets):
        features = np.ascontiguousarray(features, dtype=np.float32)
        targets = list(targets)
        if len(targets):
            features = features.reshape(len(targets), -1)
            self._ensure(features.shape[1])
            slots = np.empty(len(targets), np.int32)
            for i, t in enumerate(targets):
                if t not in self._slot:
                    if not self._free:
                        raise RuntimeError(f"feature bank is full ({self._max_tracks} tracks); raise max_tracks")
                    self._slot[t] = self._free.pop()
                slots[i] = self._slot[t]
            check(self._eng.lib.reid_bank_update(self._eng.h, self._bank, features.ctypes.data_as(C.c_void_p),
                                                 slots.ctypes.data_as(C.c_void_p), len(targets)))
        active = set(active_targets)
        gone = [t for t in self._slot if t not in active]
        if gone:
            slots = np.asarray([self._slot.pop(t) for t in gone], np.int32)
            check(self._eng.lib.reid_bank_clear(self._eng.h, self._bank, slots.ctypes.data_as(C.c_void_p), len(gone)))
            self._free.extend(int(s) for s in slots)

    def distance(self, features, targets, max_distance=None):
        """cost[len(targets), len(features)] (float64 array holding float32 values, like the reference's np.zeros fill).
        ``max_distance`` additionally applies min_cost_matching's gate on the device."""
        targets = list(targets)
        features = np.ascontiguousarray(features, dtype=np.float32)
        m = features.shape[0] if features.ndim == 2 else 0
        out = np.zeros((len(targets), m), np.float32)
        if len(targets) and m:
            self._ensure(features.shape[1])
            slots = np.asarray([self._slot[t] for t in targets], np.int32)   # KeyError for an unknown target, as the reference
            check(self._eng.lib.reid_bank_cost(self._eng.h, self._bank, slots.ctypes.data_as(C.c_void_p), len(targets),
                                               features.ctypes.data_as(C.c_void_p), m, self._metric,
                                               C.c_float(-1.0 if max_distance is None else max_distance),
                                               out.ctypes.data_as(C.c_void_p)))
        return out.astype(np.float64)

    # ------------------------------------------------------------------ frame pipeline (engine.frame_submit / _cost / _update)
    def _slots_for(self, targets, create):
        slots = np.empty(len(targets), np.int32)
        for i, t in enumerate(targets):
            if t not in self._slot:
                if not create:
                    raise KeyError(t)
                if not self._free:
                    raise RuntimeError(f"feature bank is full ({self._max_tracks} tracks); raise max_tracks")
                self._slot[t] = self._free.pop()
            slots[i] = self._slot[t]
        return slots

    def frame_distance_begin(self, slot, targets, max_distance=None, track_boxes=None, det_boxes=None):
        """Enqueue `distance` (and, given tlwh boxes, iou_matching.iou_cost) for the frame submitted with
        `engine.frame_submit(slot, crops)`, whose embeddings never leave the device in between.  Asynchronous: submit the
        next frame, then collect with `frame_distance_end`."""
        targets = list(targets)
        self._ensure(512)
        slots = self._slots_for(targets, False) if targets else None
        self._eng.frame_cost(slot, self._bank, slots, self._metric, -1.0 if max_distance is None else max_distance,
                             track_boxes, det_boxes)
        self._pending_targets = len(targets)

    def frame_distance_end(self, slot):
        """(features[m,512], cost[len(targets),m], iou_cost | None) of `frame_distance_begin`: the frame's one wait."""
        emb, cost, iou = self._eng.frame_fetch(slot)
        if cost is None:
            cost = np.zeros((self._pending_targets, emb.shape[0]), np.float32)
        return emb, cost.astype(np.float64), iou

    def frame_distance(self, slot, targets, max_distance=None, track_boxes=None, det_boxes=None):
        self.frame_distance_begin(slot, targets, max_distance, track_boxes, det_boxes)
        return self.frame_distance_end(slot)

    def frame_partial_fit(self, slot, rows, targets, active_targets):
        """`partial_fit` with features = rows `rows` of the submitted frame's embeddings (asynchronous)."""
        targets = list(targets)
        if len(targets):
            self._ensure(512)
            self._eng.frame_update(slot, self._bank, rows, self._slots_for(targets, True))
        active = set(active_targets)
        gone = [t for t in self._slot if t not in active]
        if gone:
            slots = np.asarray([self._slot.pop(t) for t in gone], np.int32)
            check(self._eng.lib.reid_bank_clear(self._eng.h, self._bank, slots.ctypes.data_as(C.c_void_p), len(gone)))
            self._free.extend(int(s) for s in slots)
