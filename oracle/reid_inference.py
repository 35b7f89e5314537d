"""ORACLE (test infrastructure, CPU only - never imported by the product path).

The evaluation script's chain (/root/reference/reid/image_reid_inference.py:238-322) as a composition of the other oracle
modules, link by link in the reference's order:
  inference_efficient :78-135 + averaging :252-253,267-268  -> oracle.seres18.forward + oracle.postproc.tta_descriptor
  diminish_camera_bias :276                                   -> oracle.postproc.diminish_camera_bias
  compute_jaccard_distance :284, clamp :286                   -> oracle.rerank.compute_jaccard_distance
  DBSCAN :299-305 (scikit-learn, as in the reference)         -> sklearn.cluster.DBSCAN
  merged_seqs * num_labels + pseudo :308, smooth_tracklets :312 -> oracle.postproc.smooth_tracklets
  evaluate_all :317                                            -> oracle.matching.evaluate_all

Pinned by tests/golden/e2e.npz, which oracle/gen_golden.py::gen_e2e made by running the reference's OWN model class and
functions through the same chain on the same seeded problem (reid_amd.synth.e2e_problem); the script file itself cannot be
imported here (onnxruntime, cv2, ultralytics ...), so the glue between the links is restated in gen_e2e and here.
"""
import numpy as np
import torch

from . import matching, postproc, rerank, seres18


def descriptors(sd, images, bs=64, arch="seres18_ibn"):
    """float32 [n,3,256,128] -> float32 [n, 512 + num_class]: normalize((d(x) + d(hflip x)) / 2)."""
    out = []
    for i in range(0, len(images), bs):
        x = torch.from_numpy(np.ascontiguousarray(images[i:i + bs], np.float32))
        e1, l1 = seres18.forward(sd, x, arch=arch)
        e2, l2 = seres18.forward(sd, torch.flip(x, dims=[3]), arch=arch)
        out.append(postproc.tta_descriptor(e1.numpy(), l1.numpy(), e2.numpy(), l2.numpy()))
    return np.concatenate(out, 0)


def evaluate_reid(sd, prob, eps=0.5, num_gallery_cams=None, la=0.05, k1=20, k2=6, cluster_fn=None, taps=None):
    """(CMC float32 [ng], mAP) for a problem dict as reid_amd.synth.e2e_problem returns it."""
    from sklearn.cluster import DBSCAN
    g = descriptors(sd, prob["g_img"])
    q = descriptors(sd, prob["q_img"])
    ng = len(g)
    merged = np.concatenate([g, q], 0)
    cams = np.concatenate([prob["gc"], prob["qc"]])
    seqs = np.concatenate([prob["gs"], prob["qs"]])
    if taps is not None:
        taps["desc"] = merged.copy()
    merged = postproc.diminish_camera_bias(merged, cams, la)
    if taps is not None:
        taps["debiased"] = merged.copy()
    dists = rerank.compute_jaccard_distance(merged, k1, k2)
    dists[dists < 0] = 0.0
    if taps is not None:
        taps["jaccard"] = dists
    n_cams = int(num_gallery_cams) if num_gallery_cams is not None else int(prob["gc"].max()) + 1
    if cluster_fn is None:
        pseudo = DBSCAN(eps=eps, min_samples=min(10, n_cams + 1), metric="precomputed", n_jobs=-1).fit_predict(dists)
    else:
        pseudo = np.asarray(cluster_fn(dists))
    if taps is not None:
        taps["pseudo_labels"] = pseudo.copy()
    num_labels = int(pseudo.max()) + 1
    merged = postproc.smooth_tracklets(merged, seqs * num_labels + pseudo, pseudo != -1)
    if taps is not None:
        taps["smoothed"] = merged.copy()
    return matching.evaluate_all(merged[ng:], prob["ql"], prob["qc"], merged[:ng], prob["gl"], prob["gc"])
