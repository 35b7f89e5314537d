"""Multi-GPU sharding of the hot path: one process per GPU.  SURVEY.md section 8(e).

The path shards with exactly one exchange step:
  * crops are independent (eval-mode BN; InstanceNorm and SE are per-sample), so a batch is split
    contiguous-by-index across ranks, every rank embeds its shard, ONE all-gather of the [n_local, D] fp32
    embeddings follows, and each rank computes its row block of the distance matrix against all of them;
  * for a fixed gallery (BASELINE config 5) the GALLERY rows are sharded instead - the reference's own
    faiss.IndexShards pattern (reid/faiss_utils.py:121-135): every rank searches its shard for all queries
    and the per-shard (distance, index) lists are merged k-way.

Transports behind the same orchestration:
  * ``RcclComm`` - the product path and the ONLY device transport (a communicator that cannot be brought up is fatal: the
    RCCL error is printed and the caller exits non-zero - there is no stand-by): collectives inside the C ABI (reid_comm_* / reid_allgather_* /
    reid_knn_gallery_sharded_dev, csrc/comm.hip, librccl over xGMI).  Everything stays in HBM: the local shard is embedded from a
    device buffer into a device buffer, gathered on the device, the row block / the merged k-NN lists are computed on the
    device.  torch.distributed is used only to hand the 128-byte communicator id from rank 0 to the others.
  * ``HostComm`` - torch.distributed (gloo) over numpy arrays, for the CPU tests of the sharding logic (world_size 2 runs
    in any container): same shard bounds, same gather order, same merge rule.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import check


def shard_bounds(n, world, rank):
    """Contiguous [lo, hi) of ``n`` items for ``rank``; the first n % world ranks hold one extra item."""
    q, r = divmod(int(n), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def round_robin(n, world, rank):
    """Indices of a frame's detections handled by ``rank`` (per-frame tracking batches, config 4)."""
    return np.arange(rank, n, world)


def frame_rows(n, world):
    """Row of detection i in a frame slot after `Engine.frame_gather` (rank i % world holds it as its (i // world)-th crop and
    contributes a block of per = ceil(n / world) rows).  Returns (rows int64[n], per)."""
    per = (int(n) + int(world) - 1) // int(world)
    i = np.arange(int(n), dtype=np.int64)
    return (i % world) * per + i // world, per


def merge_topk(d_parts, i_parts, k):
    """k-way merge of per-shard top-k lists on the host.  d_parts/i_parts: lists of [nq, k_r] arrays with GLOBAL indices.
    Returns (D float32[nq,k] ascending, I int32[nq,k]); ties -> lowest global index (the engine's rule; the device merge
    kernel of reid_knn_gallery_sharded_dev applies the same one)."""
    d = np.concatenate(d_parts, 1)
    i = np.concatenate(i_parts, 1).astype(np.int64)
    d = np.where(i < 0, np.inf, d)
    order = np.lexsort((i, d), axis=1)[:, :k]
    return np.take_along_axis(d, order, 1).astype(np.float32), np.take_along_axis(i, order, 1).astype(np.int32)


# ------------------------------------------------------------------------------------------------ device arrays
class DevArray:
    """A row-major fp32 / int32 matrix in HBM owned by an Engine (hipMalloc through the C ABI; no torch)."""

    def __init__(self, engine, shape, dtype=np.float32):
        self.engine, self.shape, self.dtype = engine, tuple(int(s) for s in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = engine.malloc(max(self.nbytes, 16))

    @classmethod
    def from_numpy(cls, engine, a):
        a = np.ascontiguousarray(a)
        d = cls(engine, a.shape, a.dtype)
        if a.nbytes:
            engine.h2d(d.ptr, a)
        return d

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if out.nbytes:
            self.engine.d2h(out, self.ptr)
        return out

    def row_ptr(self, row):
        return self.ptr + int(row) * int(np.prod(self.shape[1:])) * self.dtype.itemsize

    def free(self):
        if self.ptr:
            self.engine.free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------ transports
class RcclComm:
    """This rank's RCCL communicator behind the C ABI (csrc/comm.hip).  ``id_bytes``: the 128-byte id made by rank 0's
    ``RcclComm.unique_id()``; ``from_env`` takes rank / world from the torchrun environment and moves the id through
    torch.distributed (gloo, host plumbing only)."""

    def __init__(self, engine, rank=0, world=1, id_bytes=None):
        self.engine, self.rank, self.world = engine, int(rank), int(world)
        if world > 1 and id_bytes is None:
            raise ValueError("RcclComm: world > 1 needs the communicator id of rank 0")
        buf = (C.c_char * _ffi.COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes)) if id_bytes is not None else None
        check(engine.lib.reid_comm_init(engine.h, self.rank, self.world, buf))
        self.active = id_bytes is not None        # collectives go through librccl (false: world 1 without a communicator - local copies)

    @staticmethod
    def unique_id():
        buf = (C.c_char * _ffi.COMM_ID_BYTES)()
        check(_ffi.lib().reid_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_env(cls, engine, single_rank_communicator=False):
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run sets them.  With world 1 no communicator is
        made (every collective is a local copy) unless ``single_rank_communicator`` asks for a real 1-rank RCCL communicator
        (exercises the RCCL calls on a single GPU).  Any failure is fatal and raises on every rank that can still be told: rank 0
        sends [ok byte | 128-byte id] in ONE broadcast, so a rank 0 that could not make the id does not leave its peers waiting
        in a collective it never joins; a failed ncclCommInitRank raises ReidHipError with RCCL's message (NCCL_DEBUG=WARN is
        set by reid_comm_init, so the cause is on stderr) - the launcher then takes the job down."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        if world == 1:
            return cls(engine, 0, 1, cls.unique_id() if single_rank_communicator else None)
        import sys
        if "torch" not in sys.modules and _ffi._lib is not None and not os.environ.get("REID_ALLOW_LATE_TORCH"):
            # torch bundles its own HIP runtime and librccl; loaded after libreid_hip.so they sit BESIDE the system copies the
            # library has already bound to (two runtimes in one process: aborts at exit, undefined before).  Loaded first, both
            # bind to torch's copies.
            raise RuntimeError("RcclComm.from_env: `import torch` must come before the first reid_amd engine is created in a "
                               "multi-rank process (one HIP runtime / librccl per process); bench.py does this")
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.zeros(1 + _ffi.COMM_ID_BYTES, dtype=torch.uint8)
        err = None
        if rank == 0:
            try:
                t[1:] = torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8)
                t[0] = 1
            except Exception as e:     # noqa: BLE001 - reported to every rank through the broadcast below, then raised
                err = e
        dist.broadcast(t, src=0)
        if int(t[0]) != 1:
            raise RuntimeError("RcclComm: rank 0 could not create the communicator id (ncclGetUniqueId)%s"
                               % (": %r" % (err,) if err is not None else ""))
        return cls(engine, rank, world, bytes(t[1:].numpy().tobytes()))

    @classmethod
    def attach(cls, engine):
        """Wraps the communicator the engine's context already has (reid_comm_info): the loop-back ranks of the tests."""
        self = object.__new__(cls)
        rank, world = C.c_int(), C.c_int()
        check(engine.lib.reid_comm_info(engine.h, C.byref(rank), C.byref(world)))
        self.engine, self.rank, self.world = engine, rank.value, world.value
        self.active = True
        return self

    def close(self):
        check(self.engine.lib.reid_comm_destroy(self.engine.h))

    # ---- collectives (device pointers, enqueued on the engine's stream)
    def all_gather(self, d_send, d_recv, nbytes):
        check(self.engine.lib.reid_allgather_dev(self.engine.h, C.c_void_p(d_send), C.c_void_p(d_recv), int(nbytes)))

    def all_gather_rows(self, d_local, n_local, row_bytes, d_out):
        """Ragged row blocks -> d_out in rank order; returns the per-rank row counts."""
        counts = (C.c_int32 * self.world)()
        total = C.c_int()
        check(self.engine.lib.reid_allgather_rows_dev(self.engine.h, C.c_void_p(d_local or 0), int(n_local), int(row_bytes),
                                                      C.c_void_p(d_out), counts, C.byref(total)))
        return list(counts)

    def all_reduce(self, values, op="max"):
        """Host floats reduced over the ranks (op 'sum' | 'max'); doubles as the barrier.  Synchronises the stream."""
        v = np.atleast_1d(np.asarray(values, np.float64)).copy()
        check(self.engine.lib.reid_allreduce_f64(self.engine.h, v.ctypes.data_as(C.POINTER(C.c_double)), int(v.size),
                                                 0 if op == "sum" else 1))
        return v

    def barrier(self):
        self.all_reduce([0.0], "sum")


class HostComm:
    """torch.distributed over host arrays (gloo): the CPU stand-in used by the world_size-2 tests."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        ok = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ok else 1
        self.rank = dist.get_rank(group) if ok else 0

    def all_gather_rows(self, local):
        """numpy [n_r, ...] on every rank -> numpy [sum n_r, ...] in rank order.  One collective for the counts and one for
        the payload (padded to the largest shard); dtype preserved (int32 indices travel as int32)."""
        import torch
        local = np.ascontiguousarray(local)
        if self.world == 1:
            return local
        dist = self.dist
        cnt = torch.tensor([local.shape[0]], dtype=torch.int64)
        counts = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(counts, cnt, group=self.group)
        counts = [int(c.item()) for c in counts]
        mx = max(counts)
        pad = np.zeros((mx,) + local.shape[1:], local.dtype)
        pad[: local.shape[0]] = local
        parts = [torch.empty(pad.shape, dtype=torch.from_numpy(pad).dtype) for _ in range(self.world)]
        dist.all_gather(parts, torch.from_numpy(pad), group=self.group)
        return np.concatenate([p.numpy()[:c] for p, c in zip(parts, counts)], 0)


def _default_comm(group=None):
    return HostComm(group)


# ------------------------------------------------------------------------------------------------ crops sharded
def embed_sharded(engine, crops_u8, comm=None, group=None):
    """Embeds this rank's contiguous shard of ``crops_u8`` (uint8[N,256,128,3], identical on every rank) and all-gathers the
    embeddings.  Returns (emb_all, (lo, hi)): with an ``RcclComm`` emb_all is a DevArray [N,512] that never left HBM (only the
    local shard of crops is uploaded), with a ``HostComm`` a torch tensor."""
    comm = comm or _default_comm(group)
    n = len(crops_u8)
    lo, hi = shard_bounds(n, comm.world, comm.rank)
    if isinstance(comm, RcclComm):
        d_crops = DevArray.from_numpy(engine, np.ascontiguousarray(crops_u8[lo:hi], dtype=np.uint8))
        try:
            emb_all = embed_sharded_dev(engine, comm, d_crops.ptr, hi - lo, n)
        finally:
            engine.sync()
            d_crops.free()
        return emb_all, (lo, hi)
    import torch
    local = engine.embed_u8(crops_u8[lo:hi]) if hi > lo else np.empty((0, 512), np.float32)
    return torch.from_numpy(comm.all_gather_rows(np.ascontiguousarray(local, np.float32))), (lo, hi)


def embed_sharded_dev(engine, comm, d_crops_local, n_local, n_total, emb_all=None, d_emb_local=None):
    """Device-resident step: uint8 crops of this rank's shard (already in HBM) -> embeddings -> all-gather.  ``emb_all``
    (DevArray [n_total,512]) is allocated when not given.  Equal shards (n_total % world == 0) take ONE ncclAllGather with no
    host synchronisation; ragged shards go through the counted gather."""
    if emb_all is None:
        emb_all = DevArray(engine, (n_total, 512), np.float32)
    lo, hi = shard_bounds(n_total, comm.world, comm.rank)
    assert hi - lo == n_local, "shard of rank %d is [%d,%d), got %d crops" % (comm.rank, lo, hi, n_local)
    if comm.world == 1:
        if n_local:
            engine.embed_u8_dev(d_crops_local, n_local, emb_all.ptr)
        return emb_all
    own = d_emb_local is None
    scratch = DevArray(engine, (max(n_local, 1), 512), np.float32) if own else None
    d_local = scratch.ptr if own else d_emb_local
    if n_local:
        engine.embed_u8_dev(d_crops_local, n_local, d_local)
    if n_total % comm.world == 0:
        comm.all_gather(d_local, emb_all.ptr, n_local * 512 * 4)
    else:
        comm.all_gather_rows(d_local, n_local, 512 * 4, emb_all.ptr)
    if own:
        engine.sync()
        scratch.free()
    return emb_all


def distmat_row_block(engine, emb_all, lo, hi, metric, out=None):
    """Row block [lo:hi) of the N x N distance matrix (each rank computes only its rows).  DevArray in -> DevArray out
    (pass ``out`` to reuse a buffer); host arrays / tensors in -> numpy out."""
    if isinstance(emb_all, DevArray):
        n, d = emb_all.shape
        if out is None:
            out = DevArray(engine, (hi - lo, n), np.float32)
        if hi > lo:
            engine.distmat_dev(emb_all.row_ptr(lo), hi - lo, emb_all.ptr, n, d, metric, out.ptr)
        return out
    e = emb_all.detach().cpu().numpy() if hasattr(emb_all, "detach") else np.asarray(emb_all)
    return engine.distmat(e[lo:hi], e, metric)


# ------------------------------------------------------------------------------------------------ gallery sharded
def knn_gallery_sharded(engine, xq, xb, k, comm=None, group=None):
    """Squared-L2 k-NN with the gallery rows sharded across ranks (faiss IndexShards pattern).
    ``xq`` [nq,d] and ``xb`` [nb,d] are identical on every rank; each rank searches xb[lo:hi].  Returns the same (D, I)
    numpy pair on every rank.  With an ``RcclComm`` the local search, the exchange (distances as fp32, indices as int32)
    and the k-way merge run on the device inside reid_knn_gallery_sharded_dev; only xq and the LOCAL shard are uploaded."""
    comm = comm or _default_comm(group)
    xq = np.ascontiguousarray(xq, np.float32)
    lo, hi = shard_bounds(len(xb), comm.world, comm.rank)
    if isinstance(comm, RcclComm):
        nq, d = xq.shape
        dq = DevArray.from_numpy(engine, xq)
        db = DevArray.from_numpy(engine, np.ascontiguousarray(xb[lo:hi], np.float32))
        dD, dI = DevArray(engine, (nq, k), np.float32), DevArray(engine, (nq, k), np.int32)
        try:
            knn_gallery_sharded_dev(engine, dq.ptr, nq, db.ptr, hi - lo, lo, d, k, dD.ptr, dI.ptr, comm.world)
            return dD.numpy(), dI.numpy()
        finally:
            for a in (dq, db, dD, dI):
                a.free()
    kk = min(k, hi - lo)
    if kk > 0:
        D, I = engine.knn(xq, xb[lo:hi], kk)
        I = np.where(I >= 0, I + lo, -1).astype(np.int32)
    else:
        D, I = np.empty((len(xq), 0), np.float32), np.empty((len(xq), 0), np.int32)
    if kk < k:                       # pad so every rank contributes [nq, k]
        D = np.concatenate([D, np.full((len(xq), k - kk), np.inf, np.float32)], 1)
        I = np.concatenate([I, np.full((len(xq), k - kk), -1, np.int32)], 1)
    if comm.world == 1:
        return merge_topk([D], [I], k)
    # two collectives: fp32 distances and int32 indices (never indices bit-cast into a float payload: -1 is a NaN pattern)
    d_all = comm.all_gather_rows(D[None]).reshape(comm.world, len(xq), k)
    i_all = comm.all_gather_rows(np.ascontiguousarray(I, np.int32)[None]).reshape(comm.world, len(xq), k)
    return merge_topk(list(d_all), list(i_all), k)


def knn_gallery_sharded_dev(engine, d_xq, nq, d_xb_local, nb_local, index_base, d, k, d_D, d_I, world=None):
    """reid_knn_gallery_sharded_dev; ``world`` (the job's rank count as the caller sees it) must be what the context's
    communicator spans - a context without one would search its own shard only."""
    if world is not None:
        w = C.c_int()
        check(engine.lib.reid_comm_info(engine.h, None, C.byref(w)))
        if w.value != int(world):
            raise RuntimeError("knn_gallery_sharded_dev: the job has %d ranks but this context's communicator spans %d "
                               "(reid_comm_init missing?)" % (world, w.value))
    check(engine.lib.reid_knn_gallery_sharded_dev(engine.h, C.c_void_p(d_xq), int(nq), C.c_void_p(d_xb_local or 0), int(nb_local),
                                                  int(index_base), int(d), int(k), C.c_void_p(d_D), C.c_void_p(d_I)))
