"""ORACLE (test infrastructure, CPU only - never imported by the product path).

numpy restatement of DeepSORT's appearance metric, [external] nwojke/deep_sort `deep_sort/nn_matching.py` as vendored by
the un-pinned `yolov8_tracking`/`deep_sort_pytorch` submodules the reference drives (`.gitmodules:1-6`; parameters from
`modification_deepsort/deep_sort.yaml:3,9`).  The source is NOT under /root/reference and the reference holds no test or
fixture for it: PARITY UNPINNED - this file follows the published algorithm (Wojke et al., "Simple Online and Realtime
Tracking with a Deep Association Metric", and the public nn_matching.py): per-target sample lists truncated to the last
`budget`, cost = min over samples of (1 - cosine) or squared euclidean, and min_cost_matching's gate.
"""
import numpy as np


def pdist_sq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    a2, b2 = np.square(a).sum(axis=1), np.square(b).sum(axis=1)
    return np.clip(-2.0 * np.dot(a, b.T) + a2[:, None] + b2[None, :], 0.0, float(np.inf))


def cosine_distance(a, b, data_is_normalized=False):
    if not data_is_normalized:
        a = np.asarray(a) / np.linalg.norm(a, axis=1, keepdims=True)
        b = np.asarray(b) / np.linalg.norm(b, axis=1, keepdims=True)
    return 1.0 - np.dot(a, b.T)


def nn_euclidean_distance(x, y):
    return np.maximum(0.0, pdist_sq(x, y).min(axis=0))


def nn_cosine_distance(x, y):
    return cosine_distance(x, y).min(axis=0)


class NearestNeighborDistanceMetric:
    def __init__(self, metric, matching_threshold, budget=None):
        self._metric = {"euclidean": nn_euclidean_distance, "cosine": nn_cosine_distance}[metric]
        self.matching_threshold = matching_threshold
        self.budget = budget
        self.samples = {}

    def partial_fit(self, features, targets, active_targets):
        for feature, target in zip(features, targets):
            self.samples.setdefault(target, []).append(feature)
            if self.budget is not None:
                self.samples[target] = self.samples[target][-self.budget:]
        self.samples = {k: self.samples[k] for k in active_targets}

    def distance(self, features, targets):
        cost = np.zeros((len(targets), len(features)))
        for i, target in enumerate(targets):
            cost[i, :] = self._metric(self.samples[target], features)
        return cost


def gate(cost, max_distance):
    """linear_assignment.min_cost_matching: cost_matrix[cost_matrix > max_distance] = max_distance + 1e-5"""
    cost = np.array(cost, copy=True)
    cost[cost > max_distance] = max_distance + 1e-5
    return cost
