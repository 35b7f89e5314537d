"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl")
or gloo on CPU for tests.  SURVEY.md section 8(e).

The path shards with exactly one exchange step:
  * crops are independent (eval-mode BN; InstanceNorm and SE are per-sample), so a batch is split
    contiguous-by-index across ranks, every rank embeds its shard, ONE all-gather of the [n_local, D] fp32
    embeddings follows, and each rank computes its row block of the distance matrix against all of them;
  * for a fixed gallery (BASELINE config 5) the GALLERY rows are sharded instead - the reference's own
    faiss.IndexShards pattern (reid/faiss_utils.py:121-135): every rank searches its shard for all queries
    and the per-shard (distance, index) lists are merged k-way.
Everything here is orchestration; the compute calls go to the engine object that is passed in.
"""
import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous [lo, hi) of ``n`` items for ``rank``; the first n % world ranks hold one extra item."""
    q, r = divmod(int(n), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def round_robin(n, world, rank):
    """Indices of a frame's detections handled by ``rank`` (per-frame tracking batches, config 4)."""
    return np.arange(rank, n, world)


def _dist():
    import torch.distributed as dist
    return dist


def all_gather_rows(local, group=None):
    """All-gather of row blocks with possibly different row counts: tensor [n_r, D] on every rank ->
    tensor [sum n_r, D] in rank order (same device/dtype).  One collective for the counts (8 bytes per
    rank) and one for the payload, padded to the largest shard."""
    import torch
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    if all(c == mx for c in counts):
        out = torch.empty((mx * world,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0)


def merge_topk(d_parts, i_parts, k):
    """k-way merge of per-shard top-k lists.  d_parts/i_parts: lists of [nq, k_r] arrays with GLOBAL indices.
    Returns (D float32[nq,k] ascending, I int32[nq,k]); ties -> lowest global index (the engine's rule)."""
    d = np.concatenate(d_parts, 1)
    i = np.concatenate(i_parts, 1).astype(np.int64)
    d = np.where(i < 0, np.inf, d)
    order = np.lexsort((i, d), axis=1)[:, :k]
    return np.take_along_axis(d, order, 1).astype(np.float32), np.take_along_axis(i, order, 1).astype(np.int32)


def embed_sharded(engine, crops_u8, group=None):
    """Embeds this rank's contiguous shard of ``crops_u8`` (uint8[N,256,128,3], identical on every rank) and
    all-gathers the embeddings: returns (emb_all float32[N,512] as a torch tensor, (lo, hi) of the local shard)."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(len(crops_u8), world, rank)
    local = engine.embed_u8(crops_u8[lo:hi]) if hi > lo else np.empty((0, 512), np.float32)
    t = torch.from_numpy(np.ascontiguousarray(local))
    if dist.is_initialized() and dist.get_backend(group) == "nccl":
        t = t.cuda()
    return all_gather_rows(t, group), (lo, hi)


def distmat_row_block(engine, emb_all, lo, hi, metric):
    """Row block [lo:hi) of the N x N distance matrix (each rank computes only its rows)."""
    e = emb_all.detach().cpu().numpy() if hasattr(emb_all, "detach") else np.asarray(emb_all)
    return engine.distmat(e[lo:hi], e, metric)


def knn_gallery_sharded(engine, xq, xb, k, group=None):
    """Squared-L2 k-NN with the gallery rows sharded across ranks (faiss IndexShards pattern).
    ``xq`` [nq,d] and ``xb`` [nb,d] are identical on every rank; each rank searches xb[lo:hi] and the partial
    (D, I) lists are all-gathered ([nq,k] per rank) and merged.  Returns the same (D, I) on every rank."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(len(xb), world, rank)
    kk = min(k, hi - lo)
    if kk > 0:
        D, I = engine.knn(xq, xb[lo:hi], kk)
        I = np.where(I >= 0, I + lo, -1)
    else:
        D, I = np.empty((len(xq), 0), np.float32), np.empty((len(xq), 0), np.int32)
    if kk < k:                       # pad so every rank contributes [nq, k]
        D = np.concatenate([D, np.full((len(xq), k - kk), np.inf, np.float32)], 1)
        I = np.concatenate([I, np.full((len(xq), k - kk), -1, np.int32)], 1)
    if world == 1:
        return merge_topk([D], [I], k)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    # one payload: distances and the int32 indices bit-cast to float32
    payload = torch.from_numpy(np.concatenate([D, np.ascontiguousarray(I, dtype=np.int32).view(np.float32)], 1)).to(dev)
    allp = all_gather_rows(payload, group).cpu().numpy().reshape(world, len(xq), 2 * k)
    d_parts = [allp[r, :, :k] for r in range(world)]
    i_parts = [np.ascontiguousarray(allp[r, :, k:]).view(np.int32) for r in range(world)]
    return merge_topk(d_parts, i_parts, k)
