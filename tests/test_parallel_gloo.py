"""world_size-2 tests of the multi-GPU orchestration over gloo on CPU.  The compute engine is replaced by a
stand-in backed by the oracle (test infrastructure): what is under test is the sharding, the all-gather
ordering, ragged shards and the k-way merge - the same code that runs over RCCL with the HIP engine."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import matching
from reid_amd import parallel


class OracleEngine:
    """Stand-in with the Engine's method names; embeddings are a cheap deterministic function of the crop."""

    def embed_u8(self, crops):
        x = crops.reshape(len(crops), -1).astype(np.float32)
        proj = np.random.default_rng(0).normal(size=(64, 512)).astype(np.float32)
        return np.ascontiguousarray(x[:, :64] @ proj / 255.0)

    def distmat(self, x, y, metric):
        return matching.euclidean_dist(x, y)

    def knn(self, xq, xb, k):
        return matching.knn_l2sqr(xq, xb, k)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_crops, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = OracleEngine()
        crops = np.random.default_rng(5).integers(0, 256, (n_crops, 8, 8, 3), dtype=np.uint8)
        emb_all, (lo, hi) = parallel.embed_sharded(eng, crops)
        block = parallel.distmat_row_block(eng, emb_all, lo, hi, 0)
        xb = np.random.default_rng(6).normal(size=(101, 16)).astype(np.float32)
        xq = xb[:9] + 0.01
        D, I = parallel.knn_gallery_sharded(eng, xq, xb, 5)
        np.savez(os.path.join(tmp, "r%d.npz" % rank), emb=emb_all.numpy(), lo=lo, hi=hi, block=block, D=D, I=I)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_crops", [10, 7, 1])       # even, ragged, and fewer crops than ranks
def test_embed_allgather_and_row_blocks(tmp_path, n_crops):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_crops, str(tmp_path)), nprocs=world, join=True)
    eng = OracleEngine()
    crops = np.random.default_rng(5).integers(0, 256, (n_crops, 8, 8, 3), dtype=np.uint8)
    ref_emb = eng.embed_u8(crops)
    ref_dist = matching.euclidean_dist(ref_emb, ref_emb)
    xb = np.random.default_rng(6).normal(size=(101, 16)).astype(np.float32)
    Dr, Ir = matching.knn_l2sqr(xb[:9] + 0.01, xb, 5)
    rows = []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        np.testing.assert_array_equal(z["emb"], ref_emb)             # gather order = crop order, on every rank
        assert (int(z["lo"]), int(z["hi"])) == parallel.shard_bounds(n_crops, world, r)
        np.testing.assert_allclose(z["block"], ref_dist[int(z["lo"]):int(z["hi"])], rtol=1e-5, atol=0.1)   # self-distances sit at the sqrt(clamp) cancellation point
        np.testing.assert_array_equal(z["I"], Ir)                    # gallery-sharded k-NN == single-process k-NN
        np.testing.assert_allclose(z["D"], Dr, rtol=1e-6)
        rows.append(z["block"])
    np.testing.assert_allclose(np.concatenate(rows, 0), ref_dist, rtol=1e-5, atol=0.1)   # self-distances sit at the sqrt(clamp) cancellation point  # row blocks tile the matrix


def test_shard_bounds_and_merge():
    for n in (0, 1, 7, 8, 4096):
        for w in (1, 2, 3, 8):
            b = [parallel.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
    assert list(parallel.round_robin(7, 3, 1)) == [1, 4]
    d0, i0 = np.asarray([[0.1, 0.5, 0.9]], np.float32), np.asarray([[4, 1, 7]], np.int32)
    d1, i1 = np.asarray([[0.5, 0.6, np.inf]], np.float32), np.asarray([[0, 12, -1]], np.int32)
    D, I = parallel.merge_topk([d0, d1], [i0, i1], 4)
    assert I.tolist() == [[4, 0, 1, 12]] and D.dtype == np.float32     # tie at 0.5 -> lower global index first


def test_frame_rows_maps_detections_into_the_gathered_slot():
    """parallel.frame_rows: after reid_frame_gather rank r contributes a block of per rows holding its round-robin share in
    order; detection i must be found at row (i % world) * per + i // world, padding rows are never addressed."""
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 8, 30, 31):
            rows, per = parallel.frame_rows(n, world)
            assert per == -(-n // world) and len(rows) == n and len(set(rows.tolist())) == n
            slot = np.full(world * per, -1)
            for r in range(world):
                mine = parallel.round_robin(n, world, r)
                assert len(mine) <= per <= len(mine) + 1 or n == 0
                slot[r * per: r * per + len(mine)] = mine
            assert (slot[rows] == np.arange(n)).all()


# ----------------------------------------------------------------------------- RCCL bootstrap (no GPU: the C calls are recorded)
class _FakeLib:
    """Stands in for libreid_hip.so's communicator entry points: records what each rank hands to reid_comm_init."""

    def __init__(self, log):
        self.log = log

    def reid_comm_init(self, h, rank, world, buf):
        self.log.update(rank=rank, world=world, id=bytes(buf) if buf is not None else None)
        return 0

    def reid_comm_destroy(self, h):
        return 0


class _FakeEngine:
    def __init__(self, log):
        self.lib, self.h = _FakeLib(log), None


def _bootstrap_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    log = {}
    # rank 0's ncclGetUniqueId is replaced by a recognisable 128-byte pattern (librccl needs a GPU)
    parallel.RcclComm.unique_id = staticmethod(lambda: bytes((7 * i + 3) % 256 for i in range(128)))
    try:
        comm = parallel.RcclComm.from_env(_FakeEngine(log))
        assert (comm.rank, comm.world) == (rank, world)
        np.save(os.path.join(tmp, "id%d.npy" % rank), np.frombuffer(log["id"], np.uint8))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_rccl_bootstrap_hands_the_same_id_to_every_rank(tmp_path):
    """parallel.RcclComm.from_env under the torchrun environment: rank 0's 128-byte communicator id reaches every rank through
    torch.distributed (gloo) and goes into reid_comm_init(ctx, rank, world, id) unchanged."""
    world = 3
    mp.spawn(_bootstrap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = np.asarray([(7 * i + 3) % 256 for i in range(128)], np.uint8)
    for r in range(world):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "id%d.npy" % r)), want)


class _FailingLib(_FakeLib):
    def reid_comm_init(self, h, rank, world, buf):
        return 1 if rank == 1 else 0          # rank 1 cannot bring its communicator up


def _fatal_worker(rank, world, port, tmp, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if mode == "id":                           # rank 0 cannot even make the id (librccl missing, ncclGetUniqueId failing)
        parallel.RcclComm.unique_id = staticmethod(lambda: (_ for _ in ()).throw(RuntimeError("no librccl")))
    else:
        parallel.RcclComm.unique_id = staticmethod(lambda: bytes(128))
    parallel.check = lambda st: (_ for _ in ()).throw(parallel._ffi.ReidHipError("ncclCommInitRank -> invalid usage")) if st else None
    eng = _FakeEngine({})
    eng.lib = _FailingLib({})
    try:
        try:
            parallel.RcclComm.from_env(eng)
            verdict = "communicator"
        except Exception as e:     # noqa: BLE001 - the text is what the test looks at
            verdict = "raised: %s" % e
        open(os.path.join(tmp, "v%d" % rank), "w").write(verdict)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_a_communicator_that_cannot_be_made_is_fatal_not_rerouted(tmp_path):
    """There is no stand-by transport: a rank whose reid_comm_init fails raises with the RCCL error; when
    rank 0 cannot create the id, EVERY rank raises after the one broadcast (nobody is left waiting in a collective)."""
    world = 2
    mp.spawn(_fatal_worker, args=(world, _free_port(), str(tmp_path), "init"), nprocs=world, join=True)
    v = [open(os.path.join(str(tmp_path), "v%d" % r)).read() for r in range(world)]
    assert v[0] == "communicator" and "invalid usage" in v[1]
    assert not hasattr(parallel, "TorchComm") and not hasattr(parallel, "comm_from_env")
    mp.spawn(_fatal_worker, args=(world, _free_port(), str(tmp_path), "id"), nprocs=world, join=True)
    v = [open(os.path.join(str(tmp_path), "v%d" % r)).read() for r in range(world)]
    assert "no librccl" in v[0] and "rank 0 could not create the communicator id" in v[1]
