"""world_size 2 / 3 / 8 tests (gloo, CPU) of the DEVICE orchestration of the multi-GPU path: the same Python that runs over
RCCL on the GPUs - parallel.embed_sharded_dev / distmat_row_block / knn_gallery_sharded (RcclComm branch) and
tracking.ShardedCameraStream (round-robin shares, reid_frame_gather, frame_rows, padding rows, bank updates) - with the real
`Engine` marshalling code on top of a stand-in for libreid_hip.so: "device" memory is host memory, the collectives of the C
ABI (reid_allgather_dev, reid_allgather_rows_dev, reid_knn_gallery_sharded_dev, reid_frame_gather) are carried by
torch.distributed (gloo) following what csrc/comm.hip does step by step, and the compute entry points are cheap deterministic
functions or the oracle.  What is under test is everything ABOVE the C ABI with more than one rank: shard bounds, ragged
counts, index_base, empty shards, padding rows, gather order, the merge rule, per-rank banks staying identical.  (The C code
itself runs with several ranks in tests/test_gpu_parity.py::test_loopback_ranks_*, on the loop-back communicator.)"""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import matching
from oracle import nn_matching as onn
from reid_amd import _ffi, parallel
from reid_amd.engine import Engine
from reid_amd.tracking import ShardedCameraStream

from standin_lib import FakeLib, _embed_rows, _mem, _v, PROJ   # noqa: F401 - the stand-in library (tests/standin_lib.py)


class FakeEngine(Engine):
    """The real Engine (argument checks, marshalling, slabs) on the stand-in library."""

    def __init__(self):
        self.lib, self.h, self.device = FakeLib(), C.c_void_p(1), 0
        self.embed_dim, self.num_class = 512, 0

    def close(self):
        self.h = None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# the seeded job every rank (and the single-process reference) sees
N_CROPS = (10, 7, 3)          # equal shards at world 2, ragged, fewer crops than ranks at world 8
FRAME_COUNTS = (5, 0, 1, 8, 13, 2, 16)
TRACKS = list(range(6))


def _job_inputs():
    rng = np.random.default_rng(5)
    crops = {n: rng.integers(0, 256, (n, 256, 128, 3), dtype=np.uint8) for n in N_CROPS}
    xb = rng.normal(size=(37, 16)).astype(np.float32)
    xb[20] = xb[3]                                           # a tie across shards: lowest global index wins
    xq = np.concatenate([xb[:6] + 0.01, xb[3:4]], 0).astype(np.float32)
    frames = [[rng.integers(0, 256, (int(rng.integers(4, 9)), int(rng.integers(3, 6)), 3), dtype=np.uint8) for _ in range(n)]
              for n in FRAME_COUNTS]
    boxes = rng.uniform(0, 300, (16, 4))
    boxes[:, 2:] = rng.uniform(10, 90, (16, 2))
    seed_feats = rng.normal(size=(len(TRACKS) * 3, 512)).astype(np.float32)
    return crops, xq, xb, frames, boxes, seed_feats


def _reference():
    """One process, no sharding: what every rank must end up with."""
    crops, xq, xb, frames, boxes, seed_feats = _job_inputs()
    ref = {"emb": {n: _embed_rows(list(crops[n].reshape(n, -1))) for n in N_CROPS}}
    ref["knn"] = matching.knn_l2sqr(xq, xb, 5)
    ref["knn_big_k"] = parallel.merge_topk([matching.knn_l2sqr(xq, xb, 37)[0]], [matching.knn_l2sqr(xq, xb, 37)[1]], 40)
    metric = onn.NearestNeighborDistanceMetric("cosine", 0.15, budget=4)
    metric.partial_fit(list(seed_feats), np.repeat(TRACKS, 3), TRACKS)
    ref["frames"] = []
    for f, cr in enumerate(frames):
        n = len(cr)
        feats = _embed_rows([c.reshape(-1) for c in cr]) if n else np.empty((0, 512), np.float32)
        cost = onn.gate(metric.distance(feats, TRACKS), 0.15) if n else np.zeros((len(TRACKS), 0))
        icost = matching.diou_cost(boxes[:len(TRACKS)], boxes[:n]) if n else None
        ref["frames"].append((feats, cost, icost))
        k = min(n, len(TRACKS))
        metric.partial_fit(list(feats[:k][::-1]), TRACKS[:k], TRACKS)        # detection k-1-i -> track i
    return ref


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.RcclComm.unique_id = staticmethod(lambda: bytes(128))       # ncclGetUniqueId needs librccl + a GPU
    eng = FakeEngine()
    try:
        comm = parallel.RcclComm.from_env(eng)
        assert (comm.rank, comm.world) == (rank, world) == (eng.lib.rank, eng.lib.world)
        crops, xq, xb, frames, boxes, seed_feats = _job_inputs()
        out = {}
        # ---- crops sharded: embed_sharded (uploads the LOCAL shard only) -> embed_sharded_dev -> one gather -> row block
        for n in N_CROPS:
            emb_all, (lo, hi) = parallel.embed_sharded(eng, crops[n], comm)
            assert isinstance(emb_all, parallel.DevArray) and (lo, hi) == parallel.shard_bounds(n, world, rank)
            out["emb%d" % n] = emb_all.numpy()
            out["block%d" % n] = parallel.distmat_row_block(eng, emb_all, lo, hi, _ffi.METRIC_L2).numpy()
        # ---- gallery sharded (index_base > 0 on every rank but 0, empty shards at world 8 > some galleries)
        D, I = parallel.knn_gallery_sharded(eng, xq, xb, 5, comm)
        out["D"], out["I"] = D, I
        D2, I2 = parallel.knn_gallery_sharded(eng, xq, xb[:3], 4, comm)          # 3 gallery rows: most shards are empty
        out["D_small"], out["I_small"] = D2, I2
        D3, I3 = parallel.knn_gallery_sharded(eng, xq, xb, 40, comm)             # k > gallery: (+inf, -1) tails
        out["D_bigk"], out["I_bigk"] = D3, I3
        # ---- tracking frames: round-robin shares, frame gather, padding rows, bank updates on every rank
        stream = ShardedCameraStream(eng, comm, 0.15, budget=4)
        stream.metric.partial_fit(seed_feats, np.repeat(TRACKS, 3), TRACKS)
        stream.submit(frames[0])
        for f, cr in enumerate(frames):
            n = len(cr)
            feats, cost, icost = stream.step(n, TRACKS, boxes[:len(TRACKS)], boxes[:n], frames[f + 1] if f + 1 < len(frames) else None)
            assert feats.shape == (n, 512) and cost.shape == (len(TRACKS), n)
            assert np.isfinite(feats).all() and np.isfinite(cost).all()
            out["f%d_feats" % f], out["f%d_cost" % f] = feats, cost
            if icost is not None:
                out["f%d_icost" % f] = icost
            k = min(n, len(TRACKS))
            stream.commit(np.arange(k)[::-1], TRACKS[:k], TRACKS)
        stream.close()
        np.savez(os.path.join(tmp, "r%d.npz" % rank), **out)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_device_orchestration_with_several_ranks(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = _reference()
    _, xq, xb, frames, boxes, _ = _job_inputs()
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        blocks = []
        for n in N_CROPS:
            np.testing.assert_array_equal(z["emb%d" % n], ref["emb"][n])          # gather order = crop order, on every rank
            lo, hi = parallel.shard_bounds(n, world, r)
            want = matching.euclidean_dist(ref["emb"][n][lo:hi], ref["emb"][n])
            np.testing.assert_allclose(z["block%d" % n].reshape(hi - lo, n), want, rtol=1e-5, atol=1e-3)
        np.testing.assert_array_equal(z["I"], ref["knn"][1])                      # == single-process search, ties -> lowest row
        np.testing.assert_allclose(z["D"], ref["knn"][0], rtol=1e-6)
        Ds, Is = matching.knn_l2sqr(xq, xb[:3], 3)
        np.testing.assert_array_equal(z["I_small"][:, :3], Is)
        assert (z["I_small"][:, 3] == -1).all() and np.isinf(z["D_small"][:, 3]).all()
        np.testing.assert_array_equal(z["I_bigk"][:, :37], ref["knn_big_k"][1][:, :37])
        assert (z["I_bigk"][:, 37:] == -1).all() and np.isinf(z["D_bigk"][:, 37:]).all()
        for f, (feats, cost, icost) in enumerate(ref["frames"]):
            np.testing.assert_allclose(z["f%d_feats" % f], feats, atol=1e-7)
            np.testing.assert_allclose(z["f%d_cost" % f], cost, atol=2e-6)        # same bank on every rank, frame after frame
            if icost is not None:
                np.testing.assert_array_equal(z["f%d_icost" % f], icost)


# ----------------------------------------------------------------------------- one rank: the batching drivers of round 6
def _seeded_stream(seed, frames, lo=0, hi=9):
    rng = np.random.default_rng(seed)
    return [[rng.integers(0, 256, (int(rng.integers(4, 9)), int(rng.integers(3, 6)), 3), dtype=np.uint8) for _ in range(int(rng.integers(lo, hi)))]
            for _ in range(frames)]


def test_multi_camera_and_lookahead_streams_equal_camera_streams_on_the_stand_in():
    """Host logic of tracking.MultiCameraStream (K cameras' crops of a frame time in one slot, per-camera banks, the grouped cost
    stage, slot-wide rows for the bank update) and tracking.LookaheadCameraStream (F frames per slot, one group per frame, costs and
    updates frame by frame) against plain CameraStreams fed the same crops - on the stand-in library, whose "network" maps a crop to
    a row regardless of the pass it rides in, so everything must agree EXACTLY: features, gated costs, DIoU costs, bank contents.
    Covers cameras / frames without detections, a camera without tracks, a short last group.  (The kernels behind the same calls:
    tests/test_gpu_parity.py::test_multi_camera_batched_stream_..., ::test_lookahead_stream_....)"""
    from reid_amd.nn_matching import NearestNeighborDistanceMetric
    from reid_amd.tracking import CameraStream, LookaheadCameraStream, MultiCameraStream

    def stream(cls, *a):
        obj = cls.__new__(cls)
        obj._own, obj.eng, obj.max_dist = False, FakeEngine(), 0.3
        if cls is MultiCameraStream:
            obj.cameras = a[0]
            obj.metrics = [NearestNeighborDistanceMetric("cosine", 0.3, 5, max_tracks=32, engine=obj.eng) for _ in range(a[0])]
            obj._frame, obj._m = 0, {}
        elif cls is LookaheadCameraStream:
            obj.frames_per_pass, obj.match_stream = a[0], a[1]
            obj.eng.frame_match_stream(a[1])
            obj.metric = NearestNeighborDistanceMetric("cosine", 0.3, 5, max_tracks=32, engine=obj.eng)
            obj._group, obj._m = 0, {}
        else:
            obj.metric = NearestNeighborDistanceMetric("cosine", 0.3, 5, max_tracks=32, engine=obj.eng)
            obj._frame = 0
        return obj

    rng = np.random.default_rng(9)
    K, frames = 3, 7
    cams = [_seeded_stream(20 + c, frames) for c in range(K)]
    tracks = [list(range(4)), [7, 9], []]
    seeds = [rng.normal(size=(len(t) * 2, 512)).astype(np.float32) for t in tracks]
    boxes = rng.uniform(0, 200, (12, 4))
    boxes[:, 2:] = rng.uniform(10, 60, (12, 2))
    mc = stream(MultiCameraStream, K)
    singles = [stream(CameraStream) for _ in range(K)]
    for c in range(K):
        if tracks[c]:
            for met in (mc.metrics[c], singles[c].metric):
                met.partial_fit(seeds[c], np.repeat(tracks[c], 2), tracks[c])
    mc.submit([cams[c][0] for c in range(K)])
    for c in range(K):
        singles[c].submit(cams[c][0])
    for f in range(frames):
        nxt = f + 1 < frames
        got = mc.step(tracks, [boxes[:len(t)] for t in tracks], [boxes[:len(cams[c][f])] for c in range(K)],
                      [cams[c][f + 1] for c in range(K)] if nxt else None)
        rows, tg = [], []
        for c in range(K):
            m = len(cams[c][f])
            feats, cost, iou = singles[c].step(tracks[c], boxes[:len(tracks[c])], boxes[:m], cams[c][f + 1] if nxt else None)
            assert np.array_equal(got[c][0], feats) and got[c][1].shape == (len(tracks[c]), m)
            if tracks[c] and m:
                assert np.array_equal(got[c][1], cost) and np.array_equal(got[c][2], iou)
            k = min(m, len(tracks[c]))
            singles[c].commit(np.arange(k), tracks[c][:k], tracks[c])
            rows.append(np.arange(k))
            tg.append(tracks[c][:k])
        mc.commit(rows, tg, tracks)
    for c in range(K):
        for t in tracks[c]:
            assert mc.metrics[c].samples_count(t) == singles[c].metric.samples_count(t)
    # look-ahead groups of three frames (7 frames: 3 + 3 + 1) of camera 0; the next group handed over at the last frame (one stream)
    # or at the first one (match stream)
    for match_stream in (False, True):
        la, one = stream(LookaheadCameraStream, 3, match_stream), stream(CameraStream)
        for met in (la.metric, one.metric):
            met.partial_fit(seeds[0], np.repeat(tracks[0], 2), tracks[0])
        groups = [[0, 1, 2], [3, 4, 5], [6]]
        la.submit_group([cams[0][f] for f in groups[0]])
        assert la.handover == (0 if match_stream else 2)
        one.submit(cams[0][0])
        for gi, g in enumerate(groups):
            for j, f in enumerate(g):
                m = len(cams[0][f])
                nxt = [cams[0][x] for x in groups[gi + 1]] if (j == la.handover and gi + 1 < len(groups)) else None
                gf, gc, gio = la.step(j, tracks[0], boxes[:4], boxes[:m], nxt)
                feats, cost, iou = one.step(tracks[0], boxes[:4], boxes[:m], cams[0][f + 1] if f + 1 < frames else None)
                assert np.array_equal(gf, feats) and gc.shape == (4, m)
                if m:
                    assert np.array_equal(gc, cost) and np.array_equal(gio, iou)
                k = min(m, 4)
                la.commit(j, np.arange(k), tracks[0][:k], tracks[0])
                one.commit(np.arange(k), tracks[0][:k], tracks[0])
        for t in tracks[0]:
            assert la.metric.samples_count(t) == one.metric.samples_count(t)
