"""Generates tests/golden/*.npz by running the REFERENCE's own modules (this container only).

    python oracle/gen_golden.py            # needs /root/reference; CPU only

The reference cannot travel to the GPU box, so its outputs on seeded inputs are
committed as small fixtures; the seeded inputs/weights are regenerated in the
tests from ``reid_amd.synth`` (same seed -> same numpy arrays).

How the reference classes are made importable (nothing was denied):
  * ``torchvision`` is absent: an empty stub module is injected (the name
    ``models`` is imported at SERes18_IBN.py:3 but never used at run time).
  * ``torch.hub.load("XingangPan/IBN-Net", "resnet18_ibn_a")`` needs network:
    it is patched to return the skeleton below, written from the published
    IBN-Net architecture (resnet_ibn.py): conv1/bn1/relu/maxpool and
    layer1..4 of BasicBlock_IBN with ibn_cfg=('a','a','a',None).  Child
    registration order (conv1, bn1, relu, conv2, bn2, downsample) matters
    because SEBasicBlock slices named_children() (SERes18_IBN.py:108-114).
    IBN-Net is un-pinned by the reference (hub default branch): that boundary
    is "parity unpinned"; everything above it is the reference's own code.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------- IBN-Net skeleton [external]
class _IBN(nn.Module):
    def __init__(self, planes, ratio=0.5):
        super().__init__()
        self.half = int(planes * ratio)
        self.IN = nn.InstanceNorm2d(self.half, affine=True)
        self.BN = nn.BatchNorm2d(planes - self.half)

    def forward(self, x):
        split = torch.split(x, self.half, 1)
        return torch.cat((self.IN(split[0].contiguous()), self.BN(split[1].contiguous())), 1)


class _BasicBlockIBN(nn.Module):
    def __init__(self, inplanes, planes, ibn=None, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = _IBN(planes) if ibn == "a" else nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.IN = None
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out += residual
        return self.relu(out)


class _ResNet18IBNa(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cfg = ("a", "a", "a", None)
        self.layer1 = self._make(64, 1, cfg[0])
        self.layer2 = self._make(128, 2, cfg[1])
        self.layer3 = self._make(256, 2, cfg[2])
        self.layer4 = self._make(512, 2, cfg[3])
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, 1000)

    def _make(self, planes, stride, ibn):
        ds = None
        if stride != 1 or self.inplanes != planes:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [_BasicBlockIBN(self.inplanes, planes, ibn, stride, ds)]
        self.inplanes = planes
        layers.append(_BasicBlockIBN(planes, planes, ibn))
        return nn.Sequential(*layers)


def _install_stubs():
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tv.models)
    torch.hub.load = lambda *a, **k: _ResNet18IBNa()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "reid"))


def _sample(t):
    """Small deterministic slice of an activation for stage-level localisation."""
    t = t.detach()
    if t.dim() == 4:
        n, c, h, w = t.shape
        return t[:, :: max(1, c // 8), :: max(1, h // 8), :: max(1, w // 4)].contiguous().numpy()
    return t.numpy()


def gen_seres18():
    from reid_amd import synth
    from reid.backbones.SERes18_IBN import seres18_ibn  # the reference's own class

    for tag, seed, n, crops_fn in (("seed0", 0, 3, synth.crops_u8), ("smooth1", 1, 5, synth.smooth_crops_u8)):
        sd_np = synth.seres18_state_dict(seed)
        model = seres18_ibn(num_classes=751, loss="triplet")
        missing = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        model.eval()
        crops = crops_fn(n, seed)
        x = torch.from_numpy(crops).float().div(255.0).sub(0.5).div(0.5).permute(0, 3, 1, 2).contiguous()
        taps = {}
        hooks = []
        for name in ["bn0", "pooling0"] + [b[0] for b in synth.SERES18_BLOCKS] + ["avgpooling"]:
            hooks.append(getattr(model, name).register_forward_hook(
                lambda m, i, o, name=name: taps.__setitem__(name, o.detach().clone())))
        with torch.no_grad():
            emb, logits = model(x)
        for h in hooks:
            h.remove()
        with torch.no_grad():
            emb1, _ = model(x[:1])          # N=1 path (squeeze() also drops the batch dim)
        out = {"seed": np.int64(seed), "n": np.int64(n), "emb": emb.numpy(), "logits": logits.numpy(),
               "emb_single0": emb1.numpy()}
        for k, v in taps.items():
            out["tap_" + k] = _sample(v)
            out["mean_" + k] = np.float64(v.double().mean().item())
            out["absmean_" + k] = np.float64(v.double().abs().mean().item())
        np.savez_compressed(os.path.join(OUT, "seres18_%s.npz" % tag), **out)
        print("seres18", tag, "emb", emb.shape, "|emb| row0", float(emb[0].norm()))


def gen_matching():
    from reid_amd import synth
    from reid.losses.utils import euclidean_dist, cosine_dist
    from reid.evaluate import evaluate_all
    from modification_deepsort.iou_matching import iou

    rng = np.random.default_rng(7)
    x = rng.normal(0, 1, (37, 64)).astype(np.float32)
    y = rng.normal(0, 1, (53, 64)).astype(np.float32)
    y[5] = x[3]                      # exact duplicate -> clamp(1e-12) branch
    out = {"x": x, "y": y,
           "euclid": euclidean_dist(torch.from_numpy(x), torch.from_numpy(y)).numpy(),
           "cosine": cosine_dist(torch.from_numpy(x), torch.from_numpy(y)).numpy()}

    # retrieval: small Market-like problem incl. junk (same pid+cam), pid==-1 rows and a query with no good match
    qf, ql, qc, gf, gl, gc = synth.clustered_embeddings(40, 300, d=32, n_ids=12, n_cams=3, seed=11)
    gl[::17] = -1
    ql[7] = 999                      # no match at all -> skipped but still in the denominator
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        cmc, ap = evaluate_all(torch.from_numpy(qf), torch.from_numpy(ql), torch.from_numpy(qc),
                               torch.from_numpy(gf), torch.from_numpy(gl), torch.from_numpy(gc))
    out.update({"ev_qf": qf, "ev_ql": ql, "ev_qc": qc, "ev_gf": gf, "ev_gl": gl, "ev_gc": gc,
                "ev_cmc": cmc.numpy(), "ev_map": np.float64(ap)})

    # DIoU: the file's own demo vector (iou_matching.py:50-53) + random/edge boxes
    demo = iou(np.asarray([10, 12, 8, 9]), np.asarray([[9, 10, 9, 9], [8, 12, 9, 10], [10, 12, 9, 8]]))
    boxes = rng.uniform(0, 200, (24, 4))
    boxes[:, 2:] = rng.uniform(5, 80, (24, 2))
    cands = rng.uniform(0, 200, (31, 4))
    cands[:, 2:] = rng.uniform(5, 80, (31, 2))
    cands[0] = boxes[0]                                   # identical
    cands[1] = [boxes[1, 0] + 2, boxes[1, 1] + 2, boxes[1, 2] / 4, boxes[1, 3] / 4]   # contained
    cands[2] = [boxes[2, 0] + 500, boxes[2, 1] + 500, 10, 10]                          # disjoint
    out.update({"diou_demo": demo, "diou_boxes": boxes, "diou_cands": cands,
                "diou": np.stack([iou(b, cands) for b in boxes], 0)})
    np.savez_compressed(os.path.join(OUT, "matching.npz"), **out)
    print("matching: demo diou", demo)


def gen_factory():
    """Pins get_model_name / load_pretrained_weights behaviour (reid_model_factory.py:122-126,158-210)."""
    import io, contextlib, json, tempfile
    from pathlib import Path
    sys.path.insert(0, os.path.join(REF, "modification_tracking"))
    import reid_model_factory as rmf

    names = ["osnet_x0_25_msmt17.pt", "resnet50_market1501.pt", "swin_transformer_market.pt", "vit_duke.pt",
             "seres18_ibn.pt", "mobilenetv2_x1_4_dukemtmcreid.pt", "osnet_ain_x1_0_msmt17.pt", "unknown.pt",
             "osnet_x1_0"]
    res = {"get_model_name": {n: rmf.get_model_name(Path(n)) for n in names},
           "is_model_in_model_types": {n: rmf.is_model_in_model_types(Path(n)) for n in names},
           "get_model_url": {n: rmf.get_model_url(Path(n)) for n in names}}

    lin = nn.Sequential(nn.Linear(4, 3), nn.Linear(3, 2))
    ck = {"state_dict": {"module.0.weight": torch.ones(3, 4), "module.0.bias": torch.zeros(3),
                         "1.weight": torch.ones(5, 5), "junk": torch.ones(1)}}
    with tempfile.NamedTemporaryFile(suffix=".pt") as f:
        torch.save(ck, f.name)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rmf.load_pretrained_weights(lin, f.name)
    res["load_pretrained"] = {"w0_is_ones": bool((lin[0].weight == 1).all()), "stdout": buf.getvalue()}
    with open(os.path.join(OUT, "factory.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("factory:", res["get_model_name"])


def _timm_stub():
    """timm is absent: swin_transformer.py:12-13 imports only trunc_normal_ and Mlp from it."""
    tl = types.ModuleType("timm.models.layers")
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    tl.Mlp = type("Mlp", (nn.Module,), {"__init__": lambda self, *a, **k: nn.Module.__init__(self)})
    for n in ("timm", "timm.models"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["timm.models.layers"] = tl


def gen_swin():
    """Reference swin_t (version v1) with a stubbed timm (only trunc_normal_ / Mlp are imported, swin_transformer.py:12-13)."""
    from reid_amd import synth
    _timm_stub()
    from reid.backbones.swin_transformer import swin_t  # the reference's own class

    seed, n = 0, 2
    sd_np = synth.swin_state_dict(seed)
    model = swin_t(num_classes=751, loss="triplet")
    res = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.eval()
    x = torch.from_numpy(synth.images_f32(n, seed))
    taps = {}
    hooks = [getattr(model, name).register_forward_hook(lambda m, i, o, name=name: taps.__setitem__(name, o.detach().clone()))
             for name in ("sfe", "stage1", "stage2", "stage3", "stage4", "norm", "avgpool")]
    with torch.no_grad():
        logits, emb = model(x)          # eval returns (y, x_norm), swin_transformer.py:422-423
    for h in hooks:
        h.remove()
    out = {"seed": np.int64(seed), "n": np.int64(n), "emb": emb.numpy(), "logits": logits.numpy()}
    for k, v in taps.items():
        if v.dim() == 4:
            out["tap_" + k] = _sample(v)
        else:
            out["tap_" + k] = v[:, :: max(1, v.shape[1] // 8)].contiguous().numpy() if v.dim() == 3 else v.numpy()
        out["mean_" + k] = np.float64(v.double().mean().item())
        out["absmean_" + k] = np.float64(v.double().abs().mean().item())
    np.savez_compressed(os.path.join(OUT, "swin_seed0.npz"), **out)
    print("swin emb", emb.shape, "absmean taps", {k: round(float(out["absmean_" + k]), 3) for k in taps})


def gen_rerank():
    """k-reciprocal Jaccard re-ranking (reid/faiss_utils.py:147-244) run from the reference's own function.

    faiss (faiss-gpu 1.7.4) is absent: a stand-in module exposes only what `search_option=3` touches -
    `get_num_gpus()` and an `IndexFlatL2` doing brute-force squared-L2 search in numpy (ties by index: faiss's own tie
    order is unpinned).  Everything after the k-NN is the reference's code, unmodified.
    """
    from reid_amd import synth

    class _IndexFlatL2:
        def __init__(self, d):
            self.d, self.xb = d, np.zeros((0, d), np.float32)

        def add(self, x):
            self.xb = np.concatenate([self.xb, np.asarray(x, np.float32)], 0)

        def search(self, x, k):
            x = np.asarray(x, np.float32)
            dist = (x * x).sum(1)[:, None] + (self.xb * self.xb).sum(1)[None, :] - 2.0 * (x @ self.xb.T)
            idx = np.argsort(dist, axis=1, kind="stable")[:, :k]
            return np.take_along_axis(dist, idx, 1), idx.astype(np.int64)

    fs = types.ModuleType("faiss")
    fs.get_num_gpus = lambda: 0
    fs.METRIC_L2 = 1   # default-argument value at faiss_utils.py:57, never used on this path
    fs.IndexFlatL2 = _IndexFlatL2
    sys.modules["faiss"] = fs
    from reid.faiss_utils import compute_jaccard_distance

    out = {}
    for tag, n, d, ids, k1, k2, seed in (("a", 240, 48, 12, 20, 6, 21), ("b", 150, 32, 9, 7, 1, 22), ("c", 96, 16, 6, 5, 3, 23)):
        _, _, _, gf, gl, _ = synth.clustered_embeddings(1, n, d=d, n_ids=ids, n_cams=2, seed=seed, sigma=0.6)
        idx = _IndexFlatL2(d)
        idx.add(gf)
        _, rank = idx.search(gf, k1)
        jac = compute_jaccard_distance(torch.from_numpy(gf), k1=k1, k2=k2, print_flag=False, search_option=3)
        out.update({f"{tag}_x": gf, f"{tag}_rank": rank.astype(np.int32), f"{tag}_k": np.asarray([k1, k2]),
                    f"{tag}_jaccard": jac.astype(np.float32)})
        print("rerank", tag, n, d, "k1/k2", k1, k2, "mean jaccard", float(jac.mean()), "zeros", int((jac == 0).sum()))
    np.savez_compressed(os.path.join(OUT, "rerank.npz"), **out)


def gen_postproc():
    """Camera de-biasing from the reference's own function (reid/inference_utils.py:5-15, imports as-is)."""
    from reid.inference_utils import diminish_camera_bias
    from reid_amd import synth
    _, _, _, gf, _, gc = synth.clustered_embeddings(1, 700, d=93, n_ids=20, n_cams=4, seed=41, sigma=0.7)
    gf = gf + 0.15 * np.random.default_rng(42).normal(0, 1, (4, 93)).astype(np.float32)[gc]   # a per-camera offset to remove
    gf = (gf / np.linalg.norm(gf, axis=1, keepdims=True)).astype(np.float32)
    out = diminish_camera_bias(torch.from_numpy(gf.copy()), torch.from_numpy(gc), la=0.05).numpy()
    # smooth_tracklets (reid/inference_utils.py:18-27) from the reference function: 9 tracklets, ~30 % of the rows not valid,
    # one tracklet without any valid row (the reference's mean of nothing is caught by its bare except)
    from reid.inference_utils import smooth_tracklets
    rng = np.random.default_rng(44)
    st_x = rng.normal(0, 1, (260, 77)).astype(np.float32)
    st_seq = rng.integers(0, 9, 260).astype(np.int64) * 3 + 1
    st_valid = rng.random(260) > 0.3
    st_valid[st_seq == 7] = False
    st_out = smooth_tracklets(torch.from_numpy(st_x.copy()), torch.from_numpy(st_seq), torch.from_numpy(st_valid)).numpy()
    np.savez_compressed(os.path.join(OUT, "postproc.npz"), x=gf, cams=gc.astype(np.int32), debiased=out,
                        st_x=st_x, st_seq=st_seq.astype(np.int32), st_valid=st_valid, st_out=st_out)
    print("postproc: debias", gf.shape, "mean |delta|", float(np.abs(out - gf).mean()), "| smooth_tracklets mean |delta|",
          float(np.abs(st_out - st_x).mean()))


def gen_config1():
    """BASELINE configs[0] / SURVEY 8(c): the reference's own SERse18_IBN on 256 seeded 128x256 crops -> emb[256,512], the
    cosine distance matrix via the reference's cosine_dist ((1 - cos) / 2, reid/losses/utils.py:12-19) and its row arg-min
    with the diagonal excluded.  Two crop sets: uniform-random pixels (seed 0, the set SURVEY 8(d) names - embeddings of noise
    images are nearly parallel, so the top-2 gaps are tiny) and smooth synthetic "persons" (seed 5, realistic gaps).
    Crops and weights are regenerated in the tests from reid_amd.synth (same seeds)."""
    from reid_amd import synth
    from reid.backbones.SERes18_IBN import seres18_ibn
    from reid.losses.utils import cosine_dist

    sd_np = synth.seres18_state_dict(0)
    model = seres18_ibn(num_classes=751, loss="triplet")
    res = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.eval()
    out = {}
    for tag, crops in (("rand0", synth.crops_u8(256, 0)), ("smooth5", synth.smooth_crops_u8(256, 5))):
        embs = []
        with torch.no_grad():
            for i in range(0, 256, 64):      # reference default --bs 64 (image_reid_inference.py:144)
                x = torch.from_numpy(crops[i:i + 64]).float().div(255.0).sub(0.5).div(0.5).permute(0, 3, 1, 2).contiguous()
                embs.append(model(x)[0])
        emb = torch.cat(embs, 0)
        dist = cosine_dist(emb, emb).numpy()
        d = dist.copy()
        np.fill_diagonal(d, np.inf)
        srt = np.sort(d, axis=1)
        out.update({tag + "_emb": emb.numpy(), tag + "_cosdist": dist.astype(np.float32),
                    tag + "_argmin": d.argmin(1).astype(np.int32), tag + "_gap": (srt[:, 1] - srt[:, 0]).astype(np.float32)})
        print("config1", tag, "emb", tuple(emb.shape), "median top-2 gap", float(np.median(srt[:, 1] - srt[:, 0])),
              "min gap", float((srt[:, 1] - srt[:, 0]).min()))
    np.savez_compressed(os.path.join(OUT, "config1.npz"), **out)


def gen_swin_config():
    """BASELINE configs[2] at a size the reference finishes in seconds: its own swin_t (v1, eval) on 64 seeded 224x224 images of
    two input sets - uniform noise (seed 0) and low-frequency structure (seed 11: of the seeds 5-11 tried, the one whose smallest
    reference top-2 gap, 3.2e-6, is above the 2e-6 the GPU test asserts for the matrix, so that every row is decided; 96-d
    embeddings of random-init weights are close together) -, the Swin counterpart of gen_config1, in ONE batch
    of 64 (image_reid_inference.py:144 --bs 64) -> emb[64,96] (x_norm, swin_transformer.py:397-427), the reference's cosine_dist
    matrix ((1 - cos) / 2, reid/losses/utils.py:12-18), its row arg-min with the diagonal excluded and the top-2 gap."""
    from reid_amd import synth
    _timm_stub()
    from reid.backbones.swin_transformer import swin_t
    from reid.losses.utils import cosine_dist

    sd_np = synth.swin_state_dict(0)
    model = swin_t(num_classes=751, loss="triplet")
    res = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.eval()
    out = {}
    for tag, x in (("noise0", synth.noise_images_f32(64, 0)), ("smooth11", synth.images_f32(64, 11))):
        with torch.no_grad():
            _, emb = model(torch.from_numpy(x))
        dist = cosine_dist(emb, emb).numpy()
        d = dist.copy()
        np.fill_diagonal(d, np.inf)
        srt = np.sort(d, axis=1)
        out.update({tag + "_emb": emb.numpy(), tag + "_cosdist": dist.astype(np.float32),
                    tag + "_argmin": d.argmin(1).astype(np.int32), tag + "_gap": (srt[:, 1] - srt[:, 0]).astype(np.float32)})
        print("swin_config", tag, "emb", tuple(emb.shape), "median top-2 gap", float(np.median(srt[:, 1] - srt[:, 0])),
              "min gap", float((srt[:, 1] - srt[:, 0]).min()), "rows with gap >= 2e-6:", int(((srt[:, 1] - srt[:, 0]) >= 2e-6).sum()))
    np.savez_compressed(os.path.join(OUT, "swin_config.npz"), **out)


def gen_config5():
    """BASELINE configs[4] / SURVEY 8(c, d): the reference's evaluate_all (reid/evaluate.py:33-105) on the synthetic
    Market-1501-sized problem - qf[3368,512], gf[15913,512] around 751 centroids, 6 cameras, pid 0 distractors - giving
    CMC, mAP, the per-query AP and the per-query top-1 gallery index of the reference's own ranking
    (argsort(score)[::-1][0], evaluate.py:58-63).  sigma 0.3 is the survey's configuration (well separated identities);
    sigma 3.0 makes the ranks non-trivial (same-identity cosine ~0.1 against the ~0.18 maximum over 15 913 negatives).  Inputs are regenerated in the tests from synth.clustered_embeddings."""
    import io, contextlib
    from reid_amd import synth
    from reid.evaluate import evaluate, evaluate_all

    out = {}
    for tag, sigma in (("s03", 0.3), ("s30", 3.0)):
        qf, ql, qc, gf, gl, gc = synth.clustered_embeddings(3368, 15913, d=512, n_ids=751, n_cams=6, seed=4, sigma=sigma)
        tq, tql, tqc = torch.from_numpy(qf), torch.from_numpy(ql), torch.from_numpy(qc)
        tg, tgl, tgc = torch.from_numpy(gf), torch.from_numpy(gl), torch.from_numpy(gc)
        with contextlib.redirect_stdout(io.StringIO()):
            cmc, mAP = evaluate_all(tq, tql, tqc, tg, tgl, tgc)
        ap = np.zeros(3368, np.float64)
        top1 = np.zeros(3368, np.int32)
        first_good = np.zeros(3368, np.int32)
        for i in range(3368):
            a, c = evaluate(tq[i], tql[i], tqc[i], tg, tgl, tgc)
            ap[i] = a
            first_good[i] = -1 if c[0] == -1 else int(np.flatnonzero(c.numpy())[0])
            score = torch.mm(tg, tq[i].view(-1, 1)).squeeze(1).numpy()      # evaluate.py:58-60
            top1[i] = np.argsort(score)[::-1][0]                             # evaluate.py:62-63
        out.update({tag + "_cmc": cmc.numpy().astype(np.float32), tag + "_map": np.float64(mAP), tag + "_ap": ap,
                    tag + "_top1": top1, tag + "_first_good": first_good})
        print("config5", tag, "Rank-1 %.6f Rank-5 %.6f mAP %.6f" % (float(cmc[0]), float(cmc[4]), float(mAP)))
    np.savez_compressed(os.path.join(OUT, "config5.npz"), **out)


def gen_siblings():
    """The sibling backbones of SERse18_IBN on the same IBN-Net skeleton, from the reference's own classes:
    CARes18_IBN (reid/backbones/CARes18.py:185-281 - its blocks use TripletAttention, :148) and EMARes18_IBN
    (reid/backbones/EMA_Res18.py:118-181).  Seeded weights load with strict=True, which pins the state_dict key layout."""
    from reid_amd import synth
    from reid.backbones.CARes18 import cares18_ibn
    from reid.backbones.EMA_Res18 import emares18_ibn

    out = {}
    for tag, ctor, sd_fn in (("ca", cares18_ibn, synth.cares18_state_dict), ("ema", emares18_ibn, synth.emares18_state_dict)):
        sd_np = sd_fn(0)
        model = ctor(num_classes=751, loss="triplet")
        res = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        model.eval()
        crops = synth.smooth_crops_u8(3, 7)
        x = torch.from_numpy(crops).float().div(255.0).sub(0.5).div(0.5).permute(0, 3, 1, 2).contiguous()
        taps, hooks = {}, []
        for name in [b[0] for b in synth.SERES18_BLOCKS] + ["avgpooling"]:
            hooks.append(getattr(model, name).register_forward_hook(
                lambda m, i, o, name=name: taps.__setitem__(name, o.detach().clone())))
        with torch.no_grad():
            emb, logits = model(x)
        for h in hooks:
            h.remove()
        out.update({tag + "_emb": emb.numpy(), tag + "_logits": logits.numpy()})
        for k, v in taps.items():
            out["%s_tap_%s" % (tag, k)] = _sample(v)
            out["%s_absmean_%s" % (tag, k)] = np.float64(v.double().abs().mean().item())
        print("siblings", tag, "emb", tuple(emb.shape), "|emb| row0", float(emb[0].norm()))
    np.savez_compressed(os.path.join(OUT, "siblings.npz"), **out)


def _faiss_stub():
    """numpy stand-in for the slice of faiss that compute_jaccard_distance(search_option=3) touches (see gen_rerank)."""
    class _IndexFlatL2:
        def __init__(self, d):
            self.d, self.xb = d, np.zeros((0, d), np.float32)

        def add(self, x):
            self.xb = np.concatenate([self.xb, np.asarray(x, np.float32)], 0)

        def search(self, x, k):
            x = np.asarray(x, np.float32)
            dist = (x * x).sum(1)[:, None] + (self.xb * self.xb).sum(1)[None, :] - 2.0 * (x @ self.xb.T)
            idx = np.argsort(dist, axis=1, kind="stable")[:, :k]
            return np.take_along_axis(dist, idx, 1), idx.astype(np.int64)

    fs = types.ModuleType("faiss")
    fs.get_num_gpus = lambda: 0
    fs.METRIC_L2 = 1
    fs.IndexFlatL2 = _IndexFlatL2
    sys.modules["faiss"] = fs
    return _IndexFlatL2


def gen_e2e():
    """The evaluation script's whole chain (reid/image_reid_inference.py:238-315) on ONE seeded problem, every link the
    reference's own code: SERse18_IBN (eval) on cat(img, hflip img) in batches of 64 -> cat(normalize(emb), normalize(logits))
    (:112-123) -> (plain + mirrored) / 2, normalize (:252-253, :267-268) -> cat(gallery, query) -> diminish_camera_bias (:276)
    -> compute_jaccard_distance (:284; search_option=3 with the numpy IndexFlatL2 stand-in, faiss absent) -> clamp at 0 (:286)
    -> sklearn DBSCAN(eps=0.5, min_samples=min(10, cams + 1), precomputed) (:299-303) -> merged_seqs * num_labels + pseudo
    (:308) -> smooth_tracklets (:312) -> evaluate_all (:317).  The script itself cannot be imported (onnxruntime, cv2,
    ultralytics, datasets: SURVEY.md 8c), so the glue between the links is restated here line by line; the mirrored view is a
    plain horizontal flip (the script's strong_inference variant adds a random pad + crop), and the Market attribute
    distance (:278-283, needs the .mat file) is left out, as for the other datasets."""
    import io, contextlib
    import torch.nn.functional as F
    from sklearn.cluster import DBSCAN
    from reid_amd import synth
    _faiss_stub()
    from reid.backbones.SERes18_IBN import seres18_ibn
    from reid.inference_utils import diminish_camera_bias, smooth_tracklets
    from reid.faiss_utils import compute_jaccard_distance
    from reid.evaluate import evaluate_all

    prob = synth.e2e_problem()
    n_cams = 4
    sd_np = synth.seres18_state_dict(0)
    model = seres18_ibn(num_classes=751, loss="triplet")
    res = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.eval()

    def inference_efficient(images):            # image_reid_inference.py:78-135, bs 64 (:144)
        t1, t2 = [], []
        with torch.no_grad():
            for i in range(0, len(images), 64):
                img1 = torch.from_numpy(images[i:i + 64])
                img2 = torch.flip(img1, dims=[3])
                embeddings, outputs = model(torch.cat((img1, img2), dim=0))
                embeddings = torch.cat((F.normalize(embeddings, dim=1), F.normalize(outputs, dim=1)), dim=1)
                t1.append(embeddings[:(len(embeddings) >> 1)])
                t2.append(embeddings[(len(embeddings) >> 1):])
        return torch.cat(t1, dim=0), torch.cat(t2, dim=0)

    def ev(q, g):
        with contextlib.redirect_stdout(io.StringIO()):
            cmc, ap = evaluate_all(q, torch.from_numpy(prob["ql"]), torch.from_numpy(prob["qc"]),
                                   g, torch.from_numpy(prob["gl"]), torch.from_numpy(prob["gc"]))
        return cmc.numpy().astype(np.float32), np.float64(ap)

    g1, g2 = inference_efficient(prob["g_img"])
    gallery = F.normalize((g1 + g2) / 2.0, dim=1)
    q1, q2 = inference_efficient(prob["q_img"])
    query = F.normalize((q1 + q2) / 2.0, dim=1)
    ng = gallery.shape[0]
    merged = torch.cat((gallery, query), dim=0)
    merged_cams = torch.cat((torch.from_numpy(prob["gc"]), torch.from_numpy(prob["qc"])), dim=0)
    merged_seqs = torch.cat((torch.from_numpy(prob["gs"]), torch.from_numpy(prob["qs"])), dim=0)
    step = 6                                    # every 6th row of the big intermediates keeps the fixture small
    out = {"row_step": np.int64(step), "desc": merged.numpy()[::step].copy()}
    out["cmc_tta"], out["map_tta"] = ev(merged[ng:], merged[:ng])
    merged = diminish_camera_bias(merged, merged_cams)
    out["debiased"] = merged.numpy()[::step].copy()
    out["cmc_debiased"], out["map_debiased"] = ev(merged[ng:], merged[:ng])
    dists = compute_jaccard_distance(merged, print_flag=False, search_option=3)
    dists[dists < 0] = 0.
    out["jaccard"] = dists.astype(np.float32)[::step].copy()
    # --eps (default 0.5, :155) is a parameter of the script: take the value inside [0.45, 0.55] that lies in the middle of
    # the widest gap between neighbouring distances, so that no DBSCAN neighbourhood decision sits within float noise of it
    vals = np.unique(dists[(dists > 0.45) & (dists < 0.55)].astype(np.float64))
    gap = int(np.argmax(np.diff(vals)))
    eps = float((vals[gap] + vals[gap + 1]) / 2)
    out["eps"] = np.float64(eps)
    pseudo = DBSCAN(eps=eps, min_samples=min(10, n_cams + 1), metric="precomputed", n_jobs=-1).fit_predict(dists)
    indices_pseudo = (pseudo != -1)
    num_labels = max(pseudo) + 1
    out["pseudo_labels"] = pseudo.astype(np.int32)
    # margin of every DBSCAN decision that depends on eps: |d - eps| of the closest pair distance to the threshold
    out["eps_margin"] = np.float64(np.abs(dists - eps).min())
    merged_seqs = merged_seqs * num_labels + pseudo
    merged = smooth_tracklets(merged, merged_seqs, indices_pseudo)
    out["smoothed"] = merged.numpy()[::step].copy()
    out["cmc"], out["map"] = ev(merged[ng:], merged[:ng])
    np.savez_compressed(os.path.join(OUT, "e2e.npz"), **out)
    print("e2e: %d gallery + %d query, descriptor %d-d | Rank-1/mAP  tta %.4f/%.4f  debiased %.4f/%.4f  final %.4f/%.4f | "
          "eps %.6f: %d clusters, %d noise points, closest distance to eps %.2e"
          % (ng, query.shape[0], merged.shape[1], out["cmc_tta"][0], out["map_tta"], out["cmc_debiased"][0], out["map_debiased"],
             out["cmc"][0], out["map"], eps, num_labels, int((pseudo == -1).sum()), out["eps_margin"]))


def gen_renorm():
    """`--renorm` checkpoints (reid/image_reid_inference.py:154,180-181): the reference's seres18_ibn(renorm=True), whose BatchNorm2d
    layers are BatchRenormalization2D (batchrenorm.py:26-40, eval math :93-95), loaded strict=True from
    synth.renorm_state_dict(synth.seres18_state_dict(2)) - which pins that key layout - and run in eval mode on 4 seeded crops."""
    from reid_amd import synth
    from reid.backbones.SERes18_IBN import seres18_ibn
    sd_np = synth.renorm_state_dict(synth.seres18_state_dict(2))
    model = seres18_ibn(num_classes=751, loss="triplet", renorm=True)
    res = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.eval()
    crops = synth.smooth_crops_u8(4, 8)
    x = torch.from_numpy(crops).float().div(255.0).sub(0.5).div(0.5).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        emb, logits = model(x)
    np.savez_compressed(os.path.join(OUT, "renorm.npz"), emb=emb.numpy(), logits=logits.numpy())
    print("renorm: emb", tuple(emb.shape), "|emb| row0", float(emb[0].norm()), "renorm keys",
          sum(1 for k in sd_np if k.endswith(".gamma")))


def gen_side():
    """The optional side-information branches: SERse18_IBN.forward(x, cam) adds cam_factor * cam_bias[cam] to the BNNeck output
    (SERes18_IBN.py:269-270, cam_factor = -1 by default :198); SwinTransformer.forward(img, view_index) adds
    side_info_coeff * side_info_embedding[view] to the SFE output (swin_transformer.py:298-302, coefficient 1.5 :279) of a model
    built with camera = 4 (:288-290).  Both from the reference's own classes in eval mode."""
    from reid_amd import synth
    _timm_stub()
    from reid.backbones.SERes18_IBN import seres18_ibn
    from reid.backbones.swin_transformer import swin_t
    sd_np = synth.seres18_state_dict(3)
    model = seres18_ibn(num_classes=751, loss="triplet")
    res = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.eval()
    crops = synth.smooth_crops_u8(4, 11)
    cam = np.asarray([5, 0, 3, 3], np.int64)
    x = torch.from_numpy(crops).float().div(255.0).sub(0.5).div(0.5).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        emb, logits = model(x, torch.from_numpy(cam))
        emb0, _ = model(x)
    assert float((emb - emb0).abs().max()) > 1e-3
    sw_np = synth.swin_state_dict(4, views=4)
    sw = swin_t(num_classes=751, loss="triplet", camera=4)
    res = sw.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sw_np.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    sw.eval()
    view = np.asarray([2, 0, 3], np.int64)
    img = torch.from_numpy(synth.images_f32(3, 4))
    with torch.no_grad():
        slog, semb = sw(img, torch.from_numpy(view))
        _, semb0 = sw(img)
    assert float((semb - semb0).abs().max()) > 1e-3
    np.savez_compressed(os.path.join(OUT, "side.npz"), cam=cam, emb=emb.numpy(), logits=logits.numpy(), view=view,
                        swin_emb=semb.numpy(), swin_logits=slog.numpy())
    print("side: cam shift", float((emb - emb0).abs().max()), "view shift", float((semb - semb0).abs().max()))


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    gen_matching()
    gen_factory()
    gen_seres18()
    gen_swin()
    gen_rerank()
    gen_postproc()
    gen_config1()
    gen_config5()
    gen_siblings()
    gen_e2e()
    gen_renorm()
    gen_side()
    gen_swin_config()
