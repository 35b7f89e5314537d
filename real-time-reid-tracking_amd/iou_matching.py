"""Box-overlap cost - mirror of modification_deepsort/iou_matching.py:5-47.

The reference's ``iou`` is DIoU (IoU minus centre-distance^2 / enclosing-diagonal^2), boxes in
(top-left x, top-left y, width, height), computed in float64.  The HIP kernel keeps the reference's operation
order with FMA contraction off, so results are bit-identical to numpy's.
"""
import numpy as np

from .engine import get_engine


def iou(bbox, candidates, device=0):
    """DIoU of ``bbox`` against every row of ``candidates`` -> float64[M]."""
    return get_engine(device).diou(np.asarray(bbox, np.float64), np.asarray(candidates, np.float64))


def iou_cost(track_boxes, detection_boxes, device=0):
    """cost[t][m] = 1 - iou(track_boxes[t], detection_boxes)[m]: the whole T x M matrix of the tracker's
    iou_cost loop ([external] deep_sort iou_matching.iou_cost) in one launch.  Gate with MAX_IOU_DISTANCE 0.7
    (deep_sort.yaml:6)."""
    return get_engine(device).diou_cost(track_boxes, detection_boxes)
