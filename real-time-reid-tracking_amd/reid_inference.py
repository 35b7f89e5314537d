"""The evaluation script's chain on the device - counterpart of reid/image_reid_inference.py (SURVEY.md section 8c,
"Python harness row"), for the seres18 / cares18 / emares18 `.pt` path without side information.

Reference flow (reid/image_reid_inference.py):
  :78-135   inference_efficient   model(cat(img, flipped img)) -> cat(normalize(emb), normalize(logits)), the two halves apart
  :252-253  gallery  = normalize((plain + mirrored) / 2)          (:267-268 the same for the queries)
  :270-276  merged   = cat(gallery, query); diminish_camera_bias(merged, merged_cams)
  :284-286  dists    = compute_jaccard_distance(merged); dists[dists < 0] = 0
  :290-305  DBSCAN(eps, min_samples=min(10, num_gallery_cams + 1), metric="precomputed") -> pseudo labels
  :308-312  merged_seqs = merged_seqs * num_labels + pseudo; smooth_tracklets(merged, merged_seqs, pseudo != -1)
  :314-322  evaluate_all(query part, ..., gallery part, ...)

Here every link is one C-ABI call on device-resident data: the [ng + nq, 512 + num_class] descriptor matrix is written by
``reid_descriptor_f32_nchw_dev`` straight into its gallery / query row ranges, de-biased, re-ranked, smoothed and evaluated
in place (``reid_cam_debias_dev`` -> ``reid_rerank_jaccard_dev`` -> ``reid_smooth_tracklets_dev`` -> ``reid_rank_eval_dev``)
without a host round trip between links.  The one exception is the reference's own: DBSCAN is scikit-learn (or cuML) on the
host there (`dists.cpu().numpy()`, :297) and stays a host call here - clustering is outside the hot path -
so the Jaccard matrix is downloaded once for it; ``cluster_fn`` replaces it.  Images enter as float32 [n,3,256,128] after the
caller's transform (the script's torchvision transforms are data loading, not this path).

The Market-1501 attribute distance (:278-283, needs the dataset's .mat file) is not part of this harness.
"""
import numpy as np

from .engine import IMG_H, IMG_W, get_engine
from .parallel import DevArray


def dbscan_pseudo_labels(dists, eps=0.5, min_samples=5):
    """image_reid_inference.py:299-305 (the sklearn branch): DBSCAN on the precomputed distance matrix, on the host."""
    from sklearn.cluster import DBSCAN
    return DBSCAN(eps=eps, min_samples=min_samples, metric="precomputed", n_jobs=-1).fit_predict(dists)


def inference_efficient(engine, images, d_out, flip=True, bs=1024):
    """Descriptors of ``images`` (float32 [n,3,256,128], host) into the device rows ``d_out`` (pointer to [n, 512 + num_class]
    fp32): upload in batches of ``bs`` images, one ``reid_descriptor_f32_nchw_dev`` per batch.  Unlike the reference function
    (:78-135) the plain / mirrored halves are averaged and renormalised here already (:252-253)."""
    images = np.asarray(images)
    if images.ndim != 4 or images.shape[1:] != (3, IMG_H, IMG_W):
        raise ValueError("images must be float32 [n,3,%d,%d], got %s" % (IMG_H, IMG_W, images.shape))
    n = images.shape[0]
    width = engine.embed_dim + engine.num_class
    stage = DevArray(engine, (min(bs, max(n, 1)), 3, IMG_H, IMG_W), np.float32)
    try:
        for i in range(0, n, bs):
            part = np.ascontiguousarray(images[i:i + bs], np.float32)
            engine.h2d(stage.ptr, part)
            engine.descriptor_dev(stage.ptr, part.shape[0], flip, d_out + i * width * 4)
        engine.sync()
    finally:
        stage.free()
    return n


def evaluate_reid(gallery_images, gallery_labels, gallery_cams, gallery_seqs, query_images, query_labels, query_cams,
                  query_seqs, num_gallery_cams=None, eps=0.5, la=0.05, k1=20, k2=6, flip=True, cluster_fn=None, taps=None,
                  verbose=True, device=0, engine=None):
    """(CMC float32 [ng], mAP float) of the script's chain for one gallery / query pair; weights must be loaded on the engine
    (``Engine.load_seres18`` / ``build_model``).  ``taps`` (a dict) receives host copies of the intermediates the
    reference-generated fixture holds: "desc", "debiased", "jaccard", "pseudo_labels", "smoothed".
    ``cluster_fn(dists float32 [N,N]) -> int labels [N]`` (-1 = noise) replaces the DBSCAN call."""
    eng = engine or get_engine(device)
    gl, gc, gs = (np.ascontiguousarray(a, np.int64).reshape(-1) for a in (gallery_labels, gallery_cams, gallery_seqs))
    ql, qc, qs = (np.ascontiguousarray(a, np.int64).reshape(-1) for a in (query_labels, query_cams, query_seqs))
    ng, nq = len(gl), len(ql)
    if len(gallery_images) != ng or len(query_images) != nq or ng < 1 or nq < 1:
        raise ValueError("images and labels disagree: %d/%d gallery, %d/%d query" % (len(gallery_images), ng, len(query_images), nq))
    n, width = ng + nq, eng.embed_dim + eng.num_class
    merged = DevArray(eng, (n, width), np.float32)
    dists = DevArray(eng, (n, n), np.float32)
    try:
        inference_efficient(eng, gallery_images, merged.ptr, flip)
        inference_efficient(eng, query_images, merged.row_ptr(ng), flip)
        if taps is not None:
            taps["desc"] = merged.numpy()
        merged_cams = np.concatenate([gc, qc]).astype(np.int32)
        merged_seqs = np.concatenate([gs, qs])
        eng.cam_debias_dev(merged.ptr, merged_cams, n, width, la)
        if taps is not None:
            taps["debiased"] = merged.numpy()
        eng.rerank_jaccard_dev(merged.ptr, n, width, k1, k2, dists.ptr)     # negatives already clamped (faiss_utils.py:239-240)
        host_dists = dists.numpy()                                          # the reference's dists.cpu().numpy() (:297)
        if taps is not None:
            taps["jaccard"] = host_dists
        cams = int(num_gallery_cams) if num_gallery_cams is not None else int(gc.max()) + 1
        cluster = cluster_fn or (lambda dd: dbscan_pseudo_labels(dd, eps, min(10, cams + 1)))
        pseudo = np.asarray(cluster(host_dists)).astype(np.int64).reshape(-1)
        if pseudo.shape[0] != n:
            raise ValueError("cluster_fn returned %d labels for %d rows" % (pseudo.shape[0], n))
        num_labels = int(pseudo.max()) + 1
        if taps is not None:
            taps["pseudo_labels"] = pseudo.copy()
        seqs = (merged_seqs * num_labels + pseudo).astype(np.int32)
        eng.smooth_tracklets_dev(merged.ptr, seqs, pseudo != -1, n, width, keep=0.1)
        if taps is not None:
            taps["smoothed"] = merged.numpy()
        cmc_sum, ap, valid = eng.rank_eval_dev(merged.row_ptr(ng), ql, qc, nq, merged.ptr, gl, gc, ng, width)
    finally:
        eng.sync()
        merged.free()
        dists.free()
    total = 0.0
    for i in range(nq):              # python-float accumulation in query order, as the reference does (evaluate.py:41-50)
        if valid[i]:
            total += float(ap[i])
    cmc = cmc_sum.astype(np.float32) / nq
    mean_ap = total / nq
    if verbose:
        print('Rank@1:%f Rank@5:%f Rank@10:%f mAP:%f' % (cmc[0], cmc[min(4, ng - 1)], cmc[min(9, ng - 1)], mean_ap))
    return cmc, mean_ap
