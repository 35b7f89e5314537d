"""Pins oracle/ (the CPU restatement) against fixtures produced by the REFERENCE's own
modules (oracle/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import matching, seres18
from reid_amd import synth


def _sample(t):
    n, c, h, w = t.shape
    return t[:, :: max(1, c // 8), :: max(1, h // 8), :: max(1, w // 4)].numpy()


TAP_MAP = {"bn0": "stem", "pooling0": "pool0", "avgpooling": "gem"}


@pytest.mark.parametrize("tag,crops_fn", [("seed0", synth.crops_u8), ("smooth1", synth.smooth_crops_u8)])
def test_seres18_oracle_matches_reference(golden_dir, tag, crops_fn):
    g = np.load(os.path.join(golden_dir, "seres18_%s.npz" % tag))
    seed, n = int(g["seed"]), int(g["n"])
    sd = synth.seres18_state_dict(seed)
    crops = crops_fn(n, seed)
    taps = {}
    emb, logits = seres18.forward(sd, seres18.preprocess_u8(crops), taps)
    # eval-mode semantics; tolerance: fp32 conv summation order only
    np.testing.assert_allclose(emb.numpy(), g["emb"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=2e-4, atol=2e-3)
    cos = (emb.numpy() * g["emb"]).sum(1) / np.linalg.norm(emb.numpy(), axis=1) / np.linalg.norm(g["emb"], axis=1)
    assert (1 - cos).max() < 1e-6
    for k in g.files:
        if not k.startswith("tap_"):
            continue
        name = k[4:]
        mine = taps[TAP_MAP.get(name, name)]
        if mine.dim() == 2:
            mine = mine[:, :, None, None]
        got = _sample(mine)
        np.testing.assert_allclose(got, g[k].reshape(got.shape), rtol=2e-4, atol=2e-4, err_msg=name)
        assert abs(float(mine.double().mean()) - float(g["mean_" + name])) < 1e-4 * max(1, abs(float(g["mean_" + name])))
    # N=1 path
    emb1, _ = seres18.forward(sd, seres18.preprocess_u8(crops[:1]))
    np.testing.assert_allclose(emb1.numpy(), g["emb_single0"], rtol=2e-4, atol=2e-4)


def test_distances_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "matching.npz"))
    np.testing.assert_allclose(matching.euclidean_dist(g["x"], g["y"]), g["euclid"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(matching.cosine_dist(g["x"], g["y"]), g["cosine"], rtol=1e-5, atol=1e-6)
    assert g["euclid"][3, 5] < 1e-2          # duplicate row: clamp(1e-12).sqrt() branch region
    # rank parity
    assert (matching.euclidean_dist(g["x"], g["y"]).argmin(1) == g["euclid"].argmin(1)).all()


def test_evaluate_all_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "matching.npz"))
    cmc, ap = matching.evaluate_all(g["ev_qf"], g["ev_ql"], g["ev_qc"], g["ev_gf"], g["ev_gl"], g["ev_gc"])
    np.testing.assert_array_equal(cmc, g["ev_cmc"])
    assert abs(ap - float(g["ev_map"])) < 1e-12


def test_diou_matches_reference_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "matching.npz"))
    demo = matching.diou([10, 12, 8, 9], [[9, 10, 9, 9], [8, 12, 9, 10], [10, 12, 9, 8]])
    np.testing.assert_array_equal(demo, g["diou_demo"])
    # the reference file's own __main__ prints these (iou_matching.py:50-53)
    np.testing.assert_allclose(demo, [0.55627998, 0.62386364, 0.79691358], atol=5e-9)
    got = np.stack([matching.diou(b, g["diou_cands"]) for b in g["diou_boxes"]], 0)
    np.testing.assert_array_equal(got, g["diou"])
    np.testing.assert_array_equal(1.0 - got, matching.diou_cost(g["diou_boxes"], g["diou_cands"]))


def test_knn_contract():
    rng = np.random.default_rng(0)
    xb = rng.normal(size=(50, 16)).astype(np.float32)
    xq = xb[:7] + 0.01 * rng.normal(size=(7, 16)).astype(np.float32)
    d, i = matching.knn_l2sqr(xq, xb, 5)
    assert (i[:, 0] == np.arange(7)).all() and (np.diff(d, axis=1) >= 0).all() and i.dtype == np.int32


def test_preprocess_known_answers():
    # (i) identity when the crop is already 128x256: 2*(x/255)-1 transposed to CHW (feature_extractor.py:40-46)
    c = synth.crops_u8(2, 3)
    out = matching.preprocess(list(c))
    ref = np.transpose((c.astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5), (0, 3, 1, 2))
    np.testing.assert_array_equal(out, ref)
    np.testing.assert_array_equal(out, seres18.preprocess_u8(c).numpy())
    # (ii) half-pixel-centre bilinear 2x2 -> 4x4 ramp (hand-computed: taps 0.25/0.75, edges clamped)
    img = np.asarray([[0.0, 1.0], [2.0, 3.0]], np.float32)[:, :, None]
    got = matching.resize_bilinear(img, (4, 4))[:, :, 0]
    row = np.asarray([0.0, 0.25, 0.75, 1.0], np.float32)
    exp = np.asarray([row, row + 0.5, row + 1.5, row + 2.0], np.float32)
    np.testing.assert_allclose(got, exp, atol=1e-7)
    # (iii) downscale 4 -> 2 without antialias: samples at 0.5 and 2.5 -> mean of neighbours
    ramp = np.arange(4, dtype=np.float32)[None, :, None].repeat(4, 0)
    np.testing.assert_allclose(matching.resize_bilinear(ramp, (2, 2))[:, :, 0], [[0.5, 2.5], [0.5, 2.5]], atol=1e-7)


def test_swin_oracle_matches_reference(golden_dir):
    """oracle/swin.py against the reference's own swin_t (v1) outputs on seeded weights, 224x224, N=2 (Q14)."""
    from oracle import swin
    g = np.load(os.path.join(golden_dir, "swin_seed0.npz"))
    seed, n = int(g["seed"]), int(g["n"])
    sd = synth.swin_state_dict(seed)
    taps = {}
    emb, logits = swin.forward(sd, torch.from_numpy(synth.images_f32(n, seed)), taps)
    np.testing.assert_allclose(emb.numpy(), g["emb"], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=5e-4, atol=5e-3)
    cos = (emb.numpy() * g["emb"]).sum(1) / np.linalg.norm(emb.numpy(), axis=1) / np.linalg.norm(g["emb"], axis=1)
    assert (1 - cos).max() < 1e-6
    # reference stage modules return NCHW; the restatement keeps NHWC tokens
    for name in ("sfe", "stage1", "stage2", "stage3", "stage4"):
        mine = taps[name].permute(0, 3, 1, 2)
        np.testing.assert_allclose(_sample(mine), g["tap_" + name], rtol=5e-4, atol=5e-4, err_msg=name)
        assert abs(float(mine.double().mean()) - float(g["mean_" + name])) < 2e-4
    # masks of the shifted blocks as emitted by synth == the reference's create_mask (loaded strict=True in gen_golden)
    m = sd["stage1.layers.0.1.attention_block.fn.fn.upper_lower_mask"]
    assert np.isinf(m[0, 48]) and m[0, 27] == 0 and np.isinf(m[48, 0]) and m[48, 28] == 0


@pytest.mark.parametrize("tag,fn,seed", [("noise0", synth.noise_images_f32, 0), ("smooth11", synth.images_f32, 11)])
def test_swin_oracle_matches_reference_rank_vectors(golden_dir, tag, fn, seed):
    """oracle/swin.py against tests/golden/swin_config.npz (the reference's swin_t on 64 images, gen_golden.gen_swin_config):
    the first 12 images of each set (images are independent in eval mode) - embeddings, their block of the reference's
    (1 - cos) / 2 matrix, and the arg-min inside that block wherever the reference's own gap there exceeds the noise."""
    from oracle import matching, swin
    g = np.load(os.path.join(golden_dir, "swin_config.npz"))
    n = 12
    emb = swin.embed(synth.swin_state_dict(0), fn(64, seed)[:n])
    ref = g[tag + "_emb"][:n]
    cos = (emb * ref).sum(1) / np.linalg.norm(emb, axis=1) / np.linalg.norm(ref, axis=1)
    assert (1 - cos).max() < 1e-6
    dist = matching.cosine_dist(emb, emb)
    want = g[tag + "_cosdist"][:n, :n]
    np.testing.assert_allclose(dist, want, atol=2e-6)
    d, w = dist.copy(), want.copy()
    np.fill_diagonal(d, np.inf)
    np.fill_diagonal(w, np.inf)
    srt = np.sort(w, axis=1)
    decided = (srt[:, 1] - srt[:, 0]) >= 4e-6
    assert decided.sum() >= n // 2
    assert np.array_equal(d.argmin(1)[decided], w.argmin(1)[decided])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_rerank_oracle_matches_reference(golden_dir, tag):
    """oracle/rerank.py against the reference's compute_jaccard_distance (faiss_utils.py:147-244) run with a numpy
    stand-in for faiss.IndexFlatL2 (oracle/gen_golden.py:gen_rerank).  k1/k2 = 20/6, 7/1 (no query expansion), 5/3."""
    from oracle import rerank
    z = np.load(os.path.join(golden_dir, "rerank.npz"))
    x, k1, k2 = z[f"{tag}_x"], int(z[f"{tag}_k"][0]), int(z[f"{tag}_k"][1])
    rank = rerank.knn_l2sqr(x, k1)
    assert np.array_equal(rank, z[f"{tag}_rank"])
    got = rerank.compute_jaccard_distance(x, k1, k2)
    # identical integer steps; the softmax is torch's in the reference and numpy's here
    np.testing.assert_allclose(got, z[f"{tag}_jaccard"], rtol=0, atol=2e-6)


def test_rerank_reciprocal_sets_hand_case():
    """k_reciprocal_neigh on a 4-point line: 0 and 1 are mutual nearest neighbours, 3 is nobody's."""
    from oracle import rerank
    x = np.asarray([[0.0], [1.0], [2.2], [10.0]], np.float32)
    rank = rerank.knn_l2sqr(x, 2)
    assert rank.tolist() == [[0, 1], [1, 0], [2, 1], [3, 2]]
    assert rerank.k_reciprocal_neigh(rank, 0, 2).tolist() == [0, 1]
    assert rerank.k_reciprocal_neigh(rank, 2, 2).tolist() == [2]      # 1 does not list 2
    assert rerank.k_reciprocal_neigh(rank, 3, 2).tolist() == [3]


def test_cam_debias_oracle_matches_reference(golden_dir):
    """oracle/postproc.py:diminish_camera_bias against the reference's own function (reid/inference_utils.py:5-15)."""
    from oracle import postproc
    z = np.load(os.path.join(golden_dir, "postproc.npz"))
    got = postproc.diminish_camera_bias(z["x"], z["cams"])
    np.testing.assert_allclose(got, z["debiased"], rtol=0, atol=2e-5)     # fp32 LAPACK inverse in the reference, fp64 here
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)


def test_descriptor_oracle_hand_vectors():
    """image_reid_inference.py:123,252-253: cat of unit vectors, mean of two views, renormalise."""
    from oracle import postproc
    e = np.asarray([[3.0, 4.0]], np.float32)
    l = np.asarray([[0.0, 2.0, 0.0]], np.float32)
    np.testing.assert_allclose(postproc.descriptor(e, l), [[0.6, 0.8, 0.0, 1.0, 0.0]], atol=1e-7)
    d = postproc.tta_descriptor(e, l, np.asarray([[4.0, 3.0]], np.float32), np.asarray([[2.0, 0.0, 0.0]], np.float32))
    want = np.asarray([[0.7, 0.7, 0.5, 0.5, 0.0]]) / np.sqrt(0.49 * 2 + 0.25 * 2)
    np.testing.assert_allclose(d, want, atol=1e-6)
    assert np.isfinite(postproc.descriptor(np.zeros((1, 2), np.float32), l)).all()      # eps clamp of F.normalize


# ----------------------------------------------------------------------------- BASELINE configs[0] and [4] at full size
@pytest.mark.parametrize("tag,crops_fn,seed", [("rand0", synth.crops_u8, 0), ("smooth5", synth.smooth_crops_u8, 5)])
def test_config1_oracle_matches_reference(golden_dir, tag, crops_fn, seed):
    """256 crops -> emb -> (1 - cos) / 2 matrix -> row arg-min: the oracle against what the reference's own SERse18_IBN and
    cosine_dist produced (oracle/gen_golden.py:gen_config1)."""
    g = np.load(os.path.join(golden_dir, "config1.npz"))
    sd = synth.seres18_state_dict(0)
    emb = seres18.embed_u8(sd, crops_fn(256, seed))
    ref = g[tag + "_emb"]
    cos = (emb * ref).sum(1) / np.linalg.norm(emb, axis=1) / np.linalg.norm(ref, axis=1)
    assert (1 - cos).max() < 1e-6
    dist = matching.cosine_dist(emb, emb)
    np.testing.assert_allclose(dist, g[tag + "_cosdist"], atol=2e-6)
    d = dist.copy()
    np.fill_diagonal(d, np.inf)
    flips = np.flatnonzero(d.argmin(1) != g[tag + "_argmin"])
    # arg-min may only move where the reference's own top-2 gap is inside fp32 rounding of the distance (values ~0.01..0.5)
    assert (g[tag + "_gap"][flips] < 1e-6).all(), (flips, g[tag + "_gap"][flips])
    print("config1 %s: %d of 256 arg-mins differ, all with reference gap < 1e-6" % (tag, len(flips)))


@pytest.mark.parametrize("tag,sigma", [("s03", 0.3), ("s30", 3.0)])
def test_config5_oracle_matches_reference(golden_dir, tag, sigma):
    """Market-1501-sized retrieval (3368 x 15913 x 512): the oracle's evaluate_all against the reference's."""
    g = np.load(os.path.join(golden_dir, "config5.npz"))
    qf, ql, qc, gf, gl, gc = synth.clustered_embeddings(3368, 15913, d=512, n_ids=751, n_cams=6, seed=4, sigma=sigma)
    cmc, mean_ap = matching.evaluate_all(qf, ql, qc, gf, gl, gc)
    np.testing.assert_array_equal(np.asarray(cmc, np.float32), g[tag + "_cmc"])
    # numpy's gf @ q and torch.mm sum in different orders: a near-tie between two gallery items may swap two ranks of one query
    # (seen: 1e-9 on the mean); first-good ranks (the CMC) are exact
    assert abs(mean_ap - float(g[tag + "_map"])) < 1e-7


def test_smooth_tracklets_oracle_matches_reference(golden_dir):
    """oracle/postproc.py:smooth_tracklets against the reference's own function (reid/inference_utils.py:18-27)."""
    from oracle import postproc
    z = np.load(os.path.join(golden_dir, "postproc.npz"))
    got = postproc.smooth_tracklets(z["st_x"], z["st_seq"], z["st_valid"])
    np.testing.assert_allclose(got, z["st_out"], rtol=0, atol=2e-6)
    untouched = ~z["st_valid"]
    assert np.array_equal(got[untouched], z["st_x"][untouched]) and untouched.sum() > 40   # incl. the tracklet with no valid row


@pytest.mark.parametrize("tag,arch,sd_fn", [("ca", "cares18_ibn", synth.cares18_state_dict), ("ema", "emares18_ibn", synth.emares18_state_dict)])
def test_sibling_backbones_oracle_matches_reference(golden_dir, tag, arch, sd_fn):
    """oracle/seres18.py with arch = cares18_ibn / emares18_ibn against the reference's own CARes18_IBN / EMARes18_IBN
    (tests/golden/siblings.npz: embedding, logits and every block output, sampled as gen_golden.py sampled them)."""
    g = np.load(os.path.join(golden_dir, "siblings.npz"))
    sd = sd_fn(0)
    taps = {}
    emb, logits = seres18.forward(sd, seres18.preprocess_u8(synth.smooth_crops_u8(3, 7)), taps, arch=arch)
    np.testing.assert_allclose(emb.numpy(), g[tag + "_emb"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(logits.numpy(), g[tag + "_logits"], rtol=2e-4, atol=2e-3)
    for name in [b[0] for b in synth.SERES18_BLOCKS]:
        got = _sample(taps[name])
        np.testing.assert_allclose(got, g["%s_tap_%s" % (tag, name)], rtol=2e-4, atol=2e-4, err_msg=name)


def test_e2e_chain_oracle_matches_reference(golden_dir):
    """oracle.reid_inference (the evaluation script's chain composed from the oracle modules) against tests/golden/e2e.npz,
    which gen_golden.gen_e2e produced with the reference's own model class, diminish_camera_bias, compute_jaccard_distance,
    smooth_tracklets and evaluate_all on the same seeded problem."""
    from oracle import reid_inference
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    step = int(g["row_step"])
    prob = synth.e2e_problem()
    taps = {}
    cmc, mean_ap = reid_inference.evaluate_reid(synth.seres18_state_dict(0), prob, eps=float(g["eps"]), num_gallery_cams=4, taps=taps)
    np.testing.assert_allclose(taps["desc"][::step], g["desc"], atol=2e-5)
    np.testing.assert_allclose(taps["debiased"][::step], g["debiased"], atol=5e-5)
    np.testing.assert_allclose(taps["jaccard"][::step], g["jaccard"], atol=2e-4)
    assert (taps["pseudo_labels"] == g["pseudo_labels"]).all()
    np.testing.assert_allclose(taps["smoothed"][::step], g["smoothed"], atol=5e-5)
    np.testing.assert_array_equal(cmc, g["cmc"])
    assert abs(mean_ap - float(g["map"])) < 1e-6


def test_renorm_checkpoint_oracle_matches_reference(golden_dir):
    """tests/golden/renorm.npz: the reference's seres18_ibn(renorm=True) (BatchRenormalization2D layers, eval branch
    batchrenorm.py:93-95) on a state_dict in the --renorm layout (synth.renorm_state_dict, loaded strict=True there)."""
    g = np.load(os.path.join(golden_dir, "renorm.npz"))
    rsd = synth.renorm_state_dict(synth.seres18_state_dict(2))
    crops = synth.smooth_crops_u8(4, 8)
    emb, logits = seres18.forward(rsd, seres18.preprocess_u8(crops))
    np.testing.assert_allclose(emb.numpy(), g["emb"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=2e-4, atol=2e-3)


def test_side_information_branches_oracle_matches_reference(golden_dir):
    """tests/golden/side.npz: the reference's SERse18_IBN.forward(x, cam) (camera bias, SERes18_IBN.py:269-270) and
    SwinTransformer.forward(img, view_index) of a model built with camera=4 (swin_transformer.py:298-302)."""
    import torch
    from oracle import swin
    g = np.load(os.path.join(golden_dir, "side.npz"))
    sd = synth.seres18_state_dict(3)
    emb, logits = seres18.forward(sd, seres18.preprocess_u8(synth.smooth_crops_u8(4, 11)), cam=g["cam"])
    np.testing.assert_allclose(emb.numpy(), g["emb"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=2e-4, atol=2e-3)
    ssd = synth.swin_state_dict(4, views=4)
    semb, slog = swin.forward(ssd, torch.from_numpy(synth.images_f32(3, 4)), view_index=g["view"])
    assert np.abs(semb.numpy() - g["swin_emb"]).max() / np.abs(g["swin_emb"]).max() < 2e-4
    assert np.abs(slog.numpy() - g["swin_logits"]).max() / np.abs(g["swin_logits"]).max() < 2e-4
