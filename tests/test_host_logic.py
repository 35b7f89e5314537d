"""CPU-side tests: C-ABI surface, weight packer, plugin-surface host logic.  No GPU needed."""
import ctypes
import json
import os
import re
from pathlib import Path

import numpy as np
import pytest

from reid_amd import _ffi, synth, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as g
    g.build()
    return ctypes.CDLL(_ffi.LIB_PATH)


def test_library_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, "include", "reid_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(reid_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 35
    for sym in declared:
        assert hasattr(built_lib, sym), sym
    assert declared == set(_ffi.EXPORTS)          # the ctypes binding covers exactly the header
    # ... and the product library exports nothing else under the reid_ prefix: experiments live in libreid_hip_debug.so
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (reid_\w+)$", nm, flags=re.M))
    assert exported == declared, exported ^ declared


def test_debug_library_matches_its_header(built_lib):
    hdr = open(os.path.join(ROOT, "include", "reid_hip_debug.h")).read()
    declared = set(re.findall(r"^int\s+(reid_debug_\w+)\s*\(", hdr, flags=re.M))
    dbg = _ffi.debug_lib()
    for sym in declared:
        assert hasattr(dbg, sym), sym
    assert declared == set(_ffi.DEBUG_EXPORTS)


def test_product_library_reads_only_the_whitelisted_environment(built_lib):
    """libreid_hip.so takes TWO environment variables, both sizing knobs whose results are bit-identical by test
    (REID_SWIN_CHUNK_MAX: test_swin_embeddings_do_not_depend_on_the_pass_size; REID_KNN_WIDE_MIN:
    test_wide_knn_equals_the_fused_fp32_search_bit_for_bit); REID_CHUNK / REID_PRECISION are read by the Python host side
    (bench.py, precision.py) and passed through the C ABI.  Every switch that selects a kernel, an arithmetic form or a summation
    order is a context field that only libreid_hip_debug.so can move - a stray REID_* variable in a tracker's environment cannot
    change embeddings."""
    import subprocess
    allowed = {"REID_SWIN_CHUNK_MAX", "REID_KNN_WIDE_MIN"}
    names = set(re.findall(r"^REID_[A-Z0-9_]+$", subprocess.run(["strings", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout,
                           flags=re.M))
    names -= {n for n in names if n.startswith(("REID_ERR_", "REID_OK", "REID_K_"))}
    assert names == allowed, names ^ allowed
    csrc = os.path.join(ROOT, "real-time-reid-tracking_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".h")) or fn in ("debug.hip", "microbench.hip"):
            continue
        for var in re.findall(r'getenv\(\s*"([^"]+)"', open(os.path.join(csrc, fn)).read()):
            assert var in allowed, (fn, var)
    # the host side: names it reads, all of them sizing / arithmetic-by-name / launch plumbing
    host = set()
    for fn in ["bench.py"] + [os.path.join("real-time-reid-tracking_amd", f) for f in os.listdir(os.path.join(ROOT, "real-time-reid-tracking_amd")) if f.endswith(".py")]:
        host |= set(re.findall(r'environ(?:\.get|\.setdefault)?[\(\[]\s*"(REID_[A-Z0-9_]+)"', open(os.path.join(ROOT, fn)).read()))
    assert host <= {"REID_CHUNK", "REID_PRECISION", "REID_HIP_LIB", "REID_BENCH_COMM1", "REID_BENCH_LIMIT_SCALE", "REID_BENCH_TEST_RANK0_WATCHDOG_DELAY", "REID_ALLOW_LATE_TORCH",
                    "REID_DEBUG_SWITCHES"}, host


def test_no_gpu_means_loud_failure(built_lib):
    """The product path must fail loudly, not fall back, when it cannot run on the device."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from reid_amd.engine import Engine
    with pytest.raises(_ffi.ReidHipError):
        Engine(0)


def test_pack_seres18_layout():
    sd = synth.seres18_state_dict(0)
    blob, manifest, info = weights.pack_seres18(sd)
    assert info == {"arch": "seres18_ibn", "embed_dim": 512, "num_class": 751}
    tab = {l.split()[0]: (int(l.split()[1]), int(l.split()[2])) for l in manifest.strip().split("\n")}
    assert all(off % 4 == 0 for off, _ in tab.values())
    assert not any("seblock.bn" in k or "cam_bias" in k for k in tab)         # unused tensors dropped (Q6)
    # stem: [64][8][24] with k = r*24 + s*3 + c
    off, cnt = tab["stem.w"]
    w = blob[off:off + cnt].reshape(64, 8, 24)
    src = sd["conv0.weight"]
    assert w[5, 2, 4 * 3 + 1] == src[5, 1, 2, 4] and (w[:, 7] == 0).all() and (w[:, :, 21:] == 0).all()
    # 3x3: [Cout][R][S][Cin]
    off, cnt = tab["b21.conv1.w"]
    w = blob[off:off + cnt].reshape(128, 3, 3, 64)
    assert w[7, 2, 1, 33] == sd["basicBlock21.block_pre.conv1.weight"][7, 33, 2, 1]
    # folded BN
    off, cnt = tab["b41.n1.bn_scale"]
    g, v = sd["basicBlock41.block_pre.bn1.weight"], sd["basicBlock41.block_pre.bn1.running_var"]
    np.testing.assert_allclose(blob[off:off + cnt], g / np.sqrt(v + 1e-5), rtol=1e-6)
    assert tab["b11.n1.in_gamma"][1] == 32 and tab["b11.n1.bn_scale"][1] == 32 and tab["b41.n1.bn_scale"][1] == 512
    assert tab["b31.se.w1"][1] == 16 * 256 and tab["b42.se.w2t"][1] == 512 * 32


def test_pack_accepts_dataparallel_and_renorm_checkpoints():
    sd = synth.seres18_state_dict(1)
    base, _, _ = weights.pack_seres18(sd)
    wrapped = {"state_dict": {"module." + k: v for k, v in sd.items()}}      # image_reid_train.py:111 / factory :172-183
    b2, _, _ = weights.pack_seres18(wrapped)
    np.testing.assert_array_equal(base, b2)
    rn = dict(sd)                                                            # --renorm layout, batchrenorm.py:26-40
    for suffix, new in (("weight", "gamma"), ("bias", "beta"), ("running_mean", "running_avg_mean"),
                        ("running_var", "running_avg_var")):
        rn["bn0." + new] = rn.pop("bn0." + suffix).reshape(1, 64, 1, 1)
    b3, _, _ = weights.pack_seres18(rn)
    np.testing.assert_array_equal(base, b3)
    # the whole --renorm layout (every BatchNorm2d a BatchRenormalization2D; pinned strict=True against the reference's
    # seres18_ibn(renorm=True) in oracle/gen_golden.py::gen_renorm), also with the DataParallel prefix, and through the model object
    full = synth.renorm_state_dict(sd)
    assert "bn0.gamma" in full and full["basicBlock11.block_pre.bn1.BN.gamma"].shape == (1, 32, 1, 1) and "bnneck.weight" in full
    b4, _, _ = weights.pack_seres18({"module." + k: v for k, v in full.items()})
    np.testing.assert_array_equal(base, b4)
    from reid_amd.backbone import SERes18IBN
    m = SERes18IBN(num_classes=751)
    missing, unexpected = m.load_state_dict(full, strict=True)
    assert not missing and not unexpected
    np.testing.assert_array_equal(weights.pack_seres18(m.state_dict())[0], base)
    with pytest.raises(KeyError):
        weights.pack_seres18({"foo": np.zeros(3)})


def test_model_factory_matches_reference_fixture(golden_dir):
    from reid_amd import reid_model_factory as rmf
    ref = json.load(open(os.path.join(golden_dir, "factory.json")))
    for name, want in ref["get_model_name"].items():
        got = rmf.get_model_name(Path(name))
        if name == "seres18_ibn.pt":
            assert want is None and got == "seres18_ibn"       # the one deliberate addition
        else:
            assert got == want, name
    for name, want in ref["get_model_url"].items():
        assert rmf.get_model_url(Path(name)) == want, name
    for name, want in ref["is_model_in_model_types"].items():
        assert rmf.is_model_in_model_types(Path(name)) == want, name


def test_load_pretrained_weights_behaviour(golden_dir, tmp_path, capsys):
    import torch
    from reid_amd import reid_model_factory as rmf
    ref = json.load(open(os.path.join(golden_dir, "factory.json")))["load_pretrained"]
    lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    ck = {"state_dict": {"module.0.weight": torch.ones(3, 4), "module.0.bias": torch.zeros(3),
                         "1.weight": torch.ones(5, 5), "junk": torch.ones(1)}}
    p = tmp_path / "w.pt"
    torch.save(ck, p)
    rmf.load_pretrained_weights(lin, str(p))
    out = capsys.readouterr().out
    assert bool((lin[0].weight == 1).all()) == ref["w0_is_ones"]
    assert "discarded" in out and "['1.weight', 'junk']" in out and "['1.weight', 'junk']" in ref["stdout"]
    # nothing matches -> warning, no exception (reid_model_factory.py:194-199)
    torch.save({"zzz": torch.ones(2)}, p)
    with pytest.warns(UserWarning):
        rmf.load_pretrained_weights(lin, str(p))


def test_build_model_registry():
    from reid_amd import models
    with pytest.raises(KeyError, match="Unknown model"):
        models.build_model("resnet50", 751)
    m = models.build_model("seres18_ibn", num_classes=10, loss="triplet", pretrained=False, use_gpu=True)
    assert m.eval() is m and m.half() is m and m.to("cuda:0") is m
    sd = m.state_dict()
    assert tuple(sd["classifier.0.weight"].shape) == (10, 512) and len(sd) == len(synth.seres18_state_dict(0))
    missing, unexpected = m.load_state_dict({"module.bn0.weight": np.ones(64, np.float32), "nope": np.ones(1)}, strict=False)
    assert "nope" in unexpected and float(m.state_dict()["bn0.weight"][0]) == 1.0
    with pytest.raises(RuntimeError):
        m.to("cpu")
    with pytest.raises(NotImplementedError):
        models.build_model("seres18_ibn", 751, loss="arcface")


def test_pack_swin_layout():
    sd = synth.swin_state_dict(0)
    blob, manifest, info = weights.pack_swin(sd)
    assert info == {"arch": "swin_transformer", "embed_dim": 96, "num_class": 751}
    tab = {l.split()[0]: (int(l.split()[1]), int(l.split()[2])) for l in manifest.strip().split("\n")}
    assert all(off % 4 == 0 for off, _ in tab.values())
    assert "s1.merge.w" not in tab and not any("mask" in k for k in tab)      # unused tensors dropped
    assert len([k for k in tab if k.endswith(".qkv.w")]) == 12
    # patch merging: nn.Unfold feature order (c, kh, kw) -> (kh, kw, c)
    off, cnt = tab["s2.merge.w"]
    w = blob[off:off + cnt].reshape(192, 2, 2, 96)
    src = sd["stage2.patch_partition.linear.weight"]
    assert w[5, 1, 0, 17] == src[5, 17 * 4 + 1 * 2 + 0]
    # ConvTranspose2d(4,2,1) parity blocks: output row 2j+0 <- input rows j-1 (kernel row 3), j (kernel row 1)
    off, cnt = tab["align.t0.w"]
    w = blob[off:off + cnt].reshape(2, 2, 384, 2, 2, 768)
    src = sd["stage4_channel_align.weight"]
    assert w[0, 1, 7, 0, 1, 33] == src[33, 7, 3, 0] and w[1, 0, 7, 1, 0, 33] == src[33, 7, 0, 3]
    # the parity decomposition reproduces torch's conv_transpose2d on a small case
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    wt = rng.normal(size=(5, 3, 4, 4)).astype(np.float32)
    x = rng.normal(size=(1, 5, 4, 6)).astype(np.float32)
    ref = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(wt), None, 2, 1).numpy()
    par = weights._convt_parity(wt)
    out = np.zeros_like(ref)
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    for py in range(2):
        for px in range(2):
            for j in range(4):
                for i in range(6):
                    acc = np.zeros(3, np.float32)
                    for r in range(2):
                        for s in range(2):
                            acc += par[py, px, :, r, s, :] @ xp[0, :, j + py + r, i + px + s]
                    out[0, :, 2 * j + py, 2 * i + px] = acc
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-5)


def test_build_model_swin_registry():
    from reid_amd import models
    m = models.build_model("swin_transformer", num_classes=751, loss="triplet", pretrained=False, use_gpu=True)
    assert m.embed_dim == 96 and len(m.state_dict()) == len(synth.swin_state_dict(0))


def test_nn_matching_oracle_known_answers():
    """oracle/nn_matching.py (DeepSORT's published nn_matching, parity unpinned): hand-checked cosine / euclidean
    nearest-sample costs, budget truncation and the min_cost_matching gate."""
    from oracle import nn_matching as nm
    m = nm.NearestNeighborDistanceMetric("cosine", 0.15, budget=2)
    e = np.eye(4, dtype=np.float32)
    m.partial_fit([e[0], e[1], e[2]], [7, 7, 7], [7])          # budget 2 keeps e1, e2
    cost = m.distance(np.stack([e[0], e[1], (e[1] + e[2]) / np.sqrt(2)]).astype(np.float32), [7])
    np.testing.assert_allclose(cost[0], [1.0, 0.0, 1 - 1 / np.sqrt(2)], atol=1e-6)
    np.testing.assert_allclose(nm.gate(cost, 0.15)[0], [0.15 + 1e-5, 0.0, 0.15 + 1e-5])
    m2 = nm.NearestNeighborDistanceMetric("euclidean", 1.0)
    m2.partial_fit([e[0] * 2, e[1]], [1, 2], [1, 2])
    np.testing.assert_allclose(m2.distance(e[:2], [1, 2]), [[1.0, 5.0], [2.0, 0.0]], atol=1e-6)
    m2.partial_fit([], [], [2])                                 # target 1 is dropped
    with pytest.raises(KeyError):
        m2.distance(e[:1], [1])


def test_load_state_dict_shape_mismatch_is_loud():
    """A checkpoint tensor of another shape raises when strict (torch: "size mismatch for ..."), is reported when not strict, and
    a classifier trained on another number of identities is taken from the checkpoint (silent skips left seeded
    tensors in place)."""
    import warnings
    from reid_amd.backbone import SERes18IBN
    m = SERes18IBN(num_classes=751)
    sd = {k: np.array(v) for k, v in m.state_dict().items()}
    sd["classifier.0.weight"] = np.ones((123, 512), np.float32)          # Market -> another dataset: resized, not skipped
    m.load_state_dict(sd, strict=True)
    assert m.num_classes == 123 and m._sd["classifier.0.weight"].shape == (123, 512)
    bad = dict(sd)
    bad["bnneck.weight"] = np.ones(7, np.float32)
    with pytest.raises(RuntimeError, match="size mismatch for bnneck.weight"):
        m.load_state_dict(bad, strict=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        missing, skipped = m.load_state_dict(bad, strict=False)
    assert "bnneck.weight" in skipped and any("bnneck.weight" in str(x.message) for x in w)


def test_headers_are_plain_c(tmp_path):
    """include/reid_hip.h (the drop-in C ABI) and include/reid_hip_debug.h compile as C99 with gcc: plain pointers and sizes, no
    C++ in the signatures."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text('#include "reid_hip.h"\n#include "reid_hip_debug.h"\nint main(void) { return 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_product_library_kernels_match_the_whitelist():
    """What ships in libreid_hip.so is what the default paths and the tested switches can launch (round-5 verdict: the product library
    carried the round's experiments - 312 kernels, 36 with scratch).  The kernels are read from the library's own code objects
    (tools/so_kernels.py: AMDGPU metadata notes of the gfx950 ELFs in .hip_fatbin) and must equal tests/golden/kernels.json by NAME;
    every kernel must be free of scratch except the ones listed there with a budget - and those are (i) builds only the fp16-storage
    side mode launches (conv3x3_c64_f16, the plain 256-wide gemm_f16 tiles) and (ii) the linear-epilogue builds of gemm_f16, whose
    11-12 spilled registers sit in the ragged-tile epilogue, not in the K loop.  A kernel that appears, disappears or starts to spill
    fails here; experiment-only builds live behind `make EXPERIMENTS=1`."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import so_kernels
    lib = os.path.join(ROOT, "real-time-reid-tracking_amd", "libreid_hip.so")
    rows = so_kernels.kernels(lib)
    names = sorted(rows)
    got = {so_kernels.short(d): rows[n] for d, n in zip(so_kernels.demangle(names), names)}
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "kernels.json")))["kernels"]
    assert len(got) == len(rows), "two kernels share a demangled name"
    assert sorted(got) == sorted(want), {"new": sorted(set(got) - set(want)), "gone": sorted(set(want) - set(got))}
    over = {k: (v["scratch"], want[k]) for k, v in got.items() if v["scratch"] > want[k]}
    assert not over, over
    budgeted = sorted(k for k, v in want.items() if v)
    assert all(k.startswith(("gemm_f16_kernel<", "conv3x3_c64_f16_kernel<")) for k in budgeted), budgeted
    assert len(budgeted) <= 14
    assert os.path.getsize(lib) < 7.5e6           # 8.0 MB with the experiments of round 5, 6.5 MB without
