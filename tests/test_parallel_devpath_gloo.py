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

PROJ = np.random.default_rng(0).normal(size=(48, 512)).astype(np.float32)


def _embed_rows(x_u8):
    """The stand-in network: a fixed projection of the first 48 bytes of a crop, L2-normalised."""
    x = np.stack([np.resize(c.reshape(-1), 48) for c in x_u8]).astype(np.float32) / 255.0 - 0.5
    e = x @ PROJ
    return (e / np.linalg.norm(e, axis=1, keepdims=True)).astype(np.float32)


def _v(p):
    if p is None:
        return 0
    if isinstance(p, int):
        return p
    return p.value or 0 if hasattr(p, "value") else C.cast(p, C.c_void_p).value or 0


def _mem(ptr, nbytes, dtype=np.uint8):
    return np.frombuffer((C.c_ubyte * int(nbytes)).from_address(_v(ptr)), dtype=dtype)


class FakeLib:
    """libreid_hip.so's entry points used by the multi-rank orchestration, over host memory + gloo."""

    def __init__(self):
        self.bufs, self.rank, self.world = {}, 0, 1
        self.frame = {}          # slot -> dict(emb [m+1,512], m)
        self.out = {}            # slot -> (emb, cost, iou)
        self.banks = {}

    # ---- runtime
    def reid_malloc(self, h, nbytes, out):
        b = C.create_string_buffer(max(int(nbytes), 16))
        self.bufs[C.addressof(b)] = b
        out._obj.value = C.addressof(b)
        return 0

    def reid_free(self, h, p):
        self.bufs.pop(_v(p), None)
        return 0

    def reid_memcpy_h2d(self, h, dst, src, n):
        C.memmove(_v(dst), _v(src), int(n))
        return 0

    reid_memcpy_d2h = reid_memcpy_h2d

    def reid_ctx_sync(self, h):
        return 0

    def reid_host_alloc(self, h, nbytes, out):
        return self.reid_malloc(h, nbytes, out)

    def reid_host_free(self, h, p):
        return 0

    # ---- compute stand-ins
    def reid_embed_u8_dev(self, h, d_crops, n, d_emb, d_logits):
        crops = _mem(d_crops, n * 256 * 128 * 3).reshape(n, -1)
        _mem(d_emb, n * 2048, np.float32)[:] = _embed_rows(list(crops)).reshape(-1)
        return 0

    def reid_distmat_dev(self, h, d_x, m, d_y, n, d, metric, d_out):
        x = _mem(d_x, m * d * 4, np.float32).reshape(m, d)
        y = _mem(d_y, n * d * 4, np.float32).reshape(n, d)
        _mem(d_out, m * n * 4, np.float32)[:] = matching.euclidean_dist(x, y).reshape(-1)
        return 0

    # ---- communicator
    def reid_comm_init(self, h, rank, world, buf):
        self.rank, self.world = rank, world
        return 0

    def reid_comm_destroy(self, h):
        return 0

    def reid_comm_info(self, h, rank, world):
        if rank is not None:
            rank._obj.value = self.rank
        if world is not None:
            world._obj.value = self.world
        return 0

    def reid_allgather_dev(self, h, d_send, d_recv, nbytes):
        nbytes = int(nbytes)
        if nbytes == 0:
            return 0
        mine = torch.from_numpy(_mem(d_send, nbytes).copy())
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        _mem(d_recv, nbytes * self.world)[:] = torch.cat(parts).numpy()
        return 0

    def reid_allgather_rows_dev(self, h, d_local, n_local, row_bytes, d_out, counts, total):
        # csrc/comm.hip: counts first, then the payload padded to the largest shard, compacted in rank order
        cnt = torch.tensor([int(n_local)], dtype=torch.int32)
        allc = [torch.zeros(1, dtype=torch.int32) for _ in range(self.world)]
        dist.all_gather(allc, cnt)
        cs = [int(c) for c in allc]
        for r, c in enumerate(cs):
            counts[r] = c
        total._obj.value = sum(cs)
        mx = max(cs)
        if mx == 0:
            return 0
        pad = np.zeros(mx * row_bytes, np.uint8)
        if n_local:
            pad[: n_local * row_bytes] = _mem(d_local, n_local * row_bytes)
        parts = [torch.empty(mx * row_bytes, dtype=torch.uint8) for _ in range(self.world)]
        dist.all_gather(parts, torch.from_numpy(pad))
        out, at = _mem(d_out, sum(cs) * row_bytes), 0
        for r, c in enumerate(cs):
            out[at: at + c * row_bytes] = parts[r].numpy()[: c * row_bytes]
            at += c * row_bytes
        return 0

    def reid_knn_gallery_sharded_dev(self, h, d_xq, nq, d_xb, nb, base, d, k, d_D, d_I):
        if self.world == 1 and base > 0:
            return -3
        xq = _mem(d_xq, nq * d * 4, np.float32).reshape(nq, d)
        D = np.full((nq, k), np.inf, np.float32)
        I = np.full((nq, k), -1, np.int32)
        if nb > 0:
            xb = _mem(d_xb, nb * d * 4, np.float32).reshape(nb, d)
            kk = min(k, nb)
            Dl, Il = matching.knn_l2sqr(xq, xb, kk)
            D[:, :kk], I[:, :kk] = Dl, Il + base
        Dall = C.create_string_buffer(nq * k * 4 * self.world)
        Iall = C.create_string_buffer(nq * k * 4 * self.world)
        Dc, Ic = np.ascontiguousarray(D), np.ascontiguousarray(I)
        self.reid_allgather_dev(h, Dc.ctypes.data, C.addressof(Dall), nq * k * 4)
        self.reid_allgather_dev(h, Ic.ctypes.data, C.addressof(Iall), nq * k * 4)
        Da = np.frombuffer(Dall, np.float32).reshape(self.world, nq, k)
        Ia = np.frombuffer(Iall, np.int32).reshape(self.world, nq, k)
        Dm, Im = parallel.merge_topk(list(Da), list(Ia), k)       # the rule knn_merge_kernel implements
        _mem(d_D, nq * k * 4, np.float32)[:] = Dm.reshape(-1)
        _mem(d_I, nq * k * 4, np.int32)[:] = Im.reshape(-1)
        return 0

    # ---- feature bank + frame pipeline (csrc/bank.hip semantics)
    def reid_bank_create(self, h, max_tracks, budget, d, out):
        self.banks[1] = {"budget": budget, "rows": {}}
        out._obj.value = 1
        return 0

    def reid_bank_destroy(self, b):
        return 0

    def reid_bank_update(self, h, b, feats, slots, n):
        f = _mem(feats, n * 2048, np.float32).reshape(n, 512)
        sl = _mem(slots, n * 4, np.int32)
        bank = self.banks[1]
        for i in range(n):
            bank["rows"][int(sl[i])] = (bank["rows"].get(int(sl[i]), []) + [f[i].copy()])[-bank["budget"]:]
        return 0

    def reid_bank_clear(self, h, b, slots, n):
        for s in _mem(slots, n * 4, np.int32):
            self.banks[1]["rows"].pop(int(s), None)
        return 0

    def reid_frame_submit(self, h, slot, packed, offs, hw, n):
        emb = np.full((n + 1, 512), np.nan, np.float32)      # the spare row is stale memory until reid_frame_gather zeroes it
        if n:
            o = _mem(offs, n * 8, np.int64)
            s = _mem(hw, n * 8, np.int32).reshape(n, 2)
            crops = [_mem(_v(packed) + int(o[i]), int(s[i, 0]) * int(s[i, 1]) * 3).copy() for i in range(n)]
            emb[:n] = _embed_rows(crops)
        self.frame[slot] = {"emb": emb, "m": n}
        return 0

    def reid_frame_gather(self, h, slot, per):
        fr = self.frame[slot]
        if not (fr["m"] <= per <= fr["m"] + 1):
            return -1
        if self.world == 1 or per == 0:
            return 0
        loc = np.zeros((per, 512), np.float32)               # padding row zeroed, as comm.hip does
        loc[: fr["m"]] = fr["emb"][: fr["m"]]
        allb = C.create_string_buffer(per * 2048 * self.world)
        self.reid_allgather_dev(h, loc.ctypes.data, C.addressof(allb), per * 2048)
        fr["emb"], fr["m"] = np.frombuffer(allb, np.float32).reshape(self.world * per, 512).copy(), self.world * per
        return 0

    def reid_frame_cost(self, h, slot, bank, slots, t, metric, max_dist, tb, db, want_emb):
        fr = self.frame[slot]
        m = fr["m"]
        emb = fr["emb"][:m]
        cost = iou = None
        if bank is not None and t and m:
            sl = _mem(slots, t * 4, np.int32)
            cost = np.empty((t, m), np.float32)
            with np.errstate(invalid="ignore", divide="ignore"):
                for i in range(t):
                    cost[i] = onn.nn_cosine_distance(np.stack(self.banks[1]["rows"][int(sl[i])]), emb)
            mdist = max_dist.value if hasattr(max_dist, "value") else max_dist
            if mdist >= 0:
                cost[cost > mdist] = mdist + 1e-5
        if _v(tb) and t and m:
            iou = matching.diou_cost(_mem(tb, t * 32, np.float64).reshape(t, 4), _mem(db, m * 32, np.float64).reshape(m, 4))
        self.out[slot] = (emb.copy(), cost, iou)
        return 0

    def reid_frame_fetch(self, h, slot, emb, cost, iou):
        e, c, i = self.out.pop(slot)
        if _v(emb):
            _mem(emb, e.size * 4, np.float32)[:] = e.reshape(-1)
        if _v(cost) and c is not None:
            _mem(cost, c.size * 4, np.float32)[:] = c.reshape(-1)
        if _v(iou) and i is not None:
            _mem(iou, i.size * 8, np.float64)[:] = i.reshape(-1)
        return 0

    def reid_frame_update(self, h, slot, bank, rows, slots, n):
        r = _mem(rows, n * 4, np.int32)
        f = np.ascontiguousarray(self.frame[slot]["emb"][r])
        return self.reid_bank_update(h, bank, f.ctypes.data, slots, n)


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
