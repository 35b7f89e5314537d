"""ORACLE (test infrastructure, never shipped or measured as the product).

numpy restatements of the reference's matching-side arithmetic.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this file.  Pinned by ``oracle/gen_golden.py`` against the
reference's own modules (``tests/golden/matching_*.npz``), except where noted
"parity unpinned" (third-party arithmetic absent from /root/reference).

Follows (paths under /root/reference):
  reid/losses/utils.py:21-35                 euclidean_dist
  reid/losses/utils.py:12-18                 cosine_dist  ((1-cos)/2)
  reid/evaluate.py:33-105                    evaluate_all / evaluate / compute_mAP
  modification_deepsort/iou_matching.py:5-47 iou  (this is DIoU)
  modification_deepsort/feature_extractor.py:31-46  _preprocess
  reid/faiss_utils.py:56-118                 brute-force squared-L2 k-NN contract
      faiss-gpu 1.7.4 itself is not in /root/reference: tie order PARITY UNPINNED.
  cv2.resize(INTER_LINEAR) is not importable here: bilinear restated from
      OpenCV's published algorithm (half-pixel centres, edge clamp, no
      antialias, float32 taps): PARITY UNPINNED (identity when size matches).
"""
import numpy as np


# ------------------------------------------------------------------ distances
def euclidean_dist(x, y):
    """sqrt(clamp(|x|^2 + |y|^2 - 2 x.y^T, 1e-12)), float32 (reid/losses/utils.py:21-35)."""
    x = np.asarray(x, np.float32)
    y = np.asarray(y, np.float32)
    xx = (x * x).sum(1, dtype=np.float32)[:, None]
    yy = (y * y).sum(1, dtype=np.float32)[None, :]
    d = (xx + yy) + np.float32(-2.0) * (x @ y.T)
    return np.sqrt(np.maximum(d, np.float32(1e-12)))


def cosine_dist(x, y):
    """(1 - cos)/2, float32 (reid/losses/utils.py:12-18)."""
    x = np.asarray(x, np.float32)
    y = np.asarray(y, np.float32)
    up = x @ y.T
    down = np.sqrt((x * x).sum(1, dtype=np.float32))[:, None] * np.sqrt((y * y).sum(1, dtype=np.float32))[None, :]
    return (np.float32(1.0) - up / down) / np.float32(2.0)


def cosine_dist_deepsort(x, y):
    """1 - x^.y^ on L2-normalised rows: DeepSORT's nn_matching cosine ([external]; gates with MAX_DIST 0.15)."""
    x = np.asarray(x, np.float32)
    y = np.asarray(y, np.float32)
    xn = x / np.linalg.norm(x, axis=1, keepdims=True)
    yn = y / np.linalg.norm(y, axis=1, keepdims=True)
    return np.float32(1.0) - xn @ yn.T


def knn_l2sqr(xq, xb, k):
    """Brute-force squared-L2 k-NN: (D float32[nq,k] ascending, I int32[nq,k]); ties -> lowest index.
    Contract of faiss bfKnn / IndexFlatL2 (reid/faiss_utils.py:56-139, METRIC_L2 = squared)."""
    xq = np.asarray(xq, np.float32)
    xb = np.asarray(xb, np.float32)
    d = (xq * xq).sum(1, dtype=np.float32)[:, None] + (xb * xb).sum(1, dtype=np.float32)[None, :] \
        - np.float32(2.0) * (xq @ xb.T)
    idx = np.argsort(d, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(d, idx, 1), idx.astype(np.int32)


# ------------------------------------------------------------------ retrieval evaluation
def compute_mAP(index, good_index, junk_index):
    """reid/evaluate.py:78-105 (cmc as int32 vector, ap as python float)."""
    ap = 0.0
    cmc = np.zeros(len(index), np.int32)
    if good_index.size == 0:
        cmc[0] = -1
        return ap, cmc
    index = index[np.isin(index, junk_index, invert=True)]
    ngood = len(good_index)
    rows_good = np.argwhere(np.isin(index, good_index)).flatten()
    cmc[rows_good[0]:] = 1
    for i in range(ngood):
        d_recall = 1.0 / ngood
        precision = (i + 1) * 1.0 / (rows_good[i] + 1)
        old_precision = i * 1.0 / rows_good[i] if rows_good[i] != 0 else 1.0
        ap = ap + d_recall * (old_precision + precision) / 2
    return ap, cmc


def evaluate_one(qf, ql, qc, gf, gl, gc):
    """reid/evaluate.py:55-75: similarity gf@q, DESCENDING order, good = same pid & other cam."""
    score = (gf @ qf.reshape(-1, 1)).squeeze(1)
    index = np.argsort(score)[::-1]
    query_index = np.argwhere(gl == ql)
    camera_index = np.argwhere(gc == qc)
    good_index = np.setdiff1d(query_index, camera_index, assume_unique=True)
    junk_index1 = np.argwhere(gl == -1)
    junk_index2 = np.intersect1d(query_index, camera_index)
    junk_index = np.append(junk_index2, junk_index1)
    return compute_mAP(index, good_index, junk_index)


def evaluate_all(qf, ql, qc, gf, gl, gc):
    """reid/evaluate.py:33-52.  Divides by ALL queries, including skipped ones (:49-50)."""
    qf, gf = np.asarray(qf, np.float32), np.asarray(gf, np.float32)
    cmc = np.zeros(gf.shape[0], np.int32)
    ap = 0.0
    for i in range(qf.shape[0]):
        ap_i, cmc_i = evaluate_one(qf[i], ql[i], qc[i], gf, gl, gc)
        if cmc_i[0] == -1:
            continue
        cmc = cmc + cmc_i
        ap += ap_i
    return cmc.astype(np.float32) / qf.shape[0], ap / qf.shape[0]


# ------------------------------------------------------------------ box cost
def diou(bbox, candidates):
    """modification_deepsort/iou_matching.py:5-47: IoU - centre_dist^2 / enclosing_diag^2, tlwh boxes, float64."""
    bbox = np.asarray(bbox, np.float64)
    cand = np.asarray(candidates, np.float64)
    b_tl, b_br = bbox[:2], bbox[:2] + bbox[2:]
    c_tl, c_br = cand[:, :2], cand[:, :2] + cand[:, 2:]
    # the reference builds centres as (y, x) for both; only the squared distance is used
    bc = np.asarray([(b_tl[1] + b_br[1]) / 2, (b_tl[0] + b_br[0]) / 2])
    cc = np.stack([(c_tl[:, 1] + c_br[:, 1]) / 2, (c_tl[:, 0] + c_br[:, 0]) / 2], 1)
    d = ((bc - cc) ** 2).sum(1)
    o_tl = np.minimum(b_tl, c_tl)
    o_br = np.maximum(b_br, c_br)
    rou = ((o_tl - o_br) ** 2).sum(1)
    wh = np.maximum(0.0, np.minimum(b_br, c_br) - np.maximum(b_tl, c_tl))
    inter = wh[:, 0] * wh[:, 1]
    iou = inter / (bbox[2] * bbox[3] + cand[:, 2] * cand[:, 3] - inter)
    return iou - d / rou


def diou_cost(tracks, dets):
    """1 - DIoU for every (track, detection) pair ([external] deep_sort iou_cost loop over iou())."""
    return np.stack([1.0 - diou(t, np.asarray(dets, np.float64)) for t in np.asarray(tracks, np.float64)], 0)


# ------------------------------------------------------------------ crop preprocessing
def _lin_taps(dst, src):
    scale = src / dst
    ofs = np.empty(dst, np.int64)
    w1 = np.empty(dst, np.float32)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= src - 1:
            s, f = src - 1, np.float32(0)
        ofs[d], w1[d] = s, f
    return ofs, w1


def resize_bilinear(img, size):
    """float32[h,w,c] -> float32[H,W,c], size=(W,H) as cv2.resize(INTER_LINEAR).  PARITY UNPINNED (no cv2)."""
    W, H = size
    h, w = img.shape[:2]
    img = np.asarray(img, np.float32)
    if (h, w) == (H, W):
        return img.copy()
    xo, xw = _lin_taps(W, w)
    yo, yw = _lin_taps(H, h)
    x1 = np.minimum(xo + 1, w - 1)
    y1 = np.minimum(yo + 1, h - 1)
    xw = xw[None, :, None]
    rows = img[:, xo] * (np.float32(1) - xw) + img[:, x1] * xw          # horizontal pass
    yw = yw[:, None, None]
    return (rows[yo] * (np.float32(1) - yw) + rows[y1] * yw).astype(np.float32)


def preprocess(im_crops, size=(128, 256)):
    """feature_extractor.py:31-46: astype(float32)/255 -> resize -> ToTensor (HWC->CHW) -> (x-0.5)/0.5."""
    out = []
    for im in im_crops:
        x = resize_bilinear(np.asarray(im).astype(np.float32) / np.float32(255.0), size)
        x = (x - np.float32(0.5)) / np.float32(0.5)
        out.append(np.transpose(x, (2, 0, 1)))
    return np.stack(out, 0).astype(np.float32)
