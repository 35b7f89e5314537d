"""Stand-in for libreid_hip.so used by the CPU (gloo) tests of everything ABOVE the C ABI with more than one rank
(tests/test_parallel_devpath_gloo.py, tests/test_bench_launch.py): "device" memory is host memory, the collectives of the C ABI
(reid_allgather_dev, reid_allgather_rows_dev, reid_knn_gallery_sharded_dev, reid_frame_gather, reid_allreduce_f64) are carried by
torch.distributed (gloo) following what csrc/comm.hip does step by step, and the compute entry points are cheap deterministic
functions or the oracle.  Test infrastructure only: the product never loads it (reid_amd._ffi.lib() opens the real library or
raises)."""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from oracle import matching
from oracle import nn_matching as onn
from reid_amd import parallel

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
        key = len(self.banks) + 1                            # (several banks per context: one per camera of a MultiCameraStream)
        self.banks[key] = {"budget": budget, "rows": {}}
        out._obj.value = key
        return 0

    def reid_bank_destroy(self, b):
        return 0

    def _bank(self, b):
        return self.banks[_v(b) or 1]

    def reid_bank_update(self, h, b, feats, slots, n):
        f = _mem(feats, n * 2048, np.float32).reshape(n, 512)
        sl = _mem(slots, n * 4, np.int32)
        bank = self._bank(b)
        for i in range(n):
            bank["rows"][int(sl[i])] = (bank["rows"].get(int(sl[i]), []) + [f[i].copy()])[-bank["budget"]:]
        return 0

    def reid_bank_clear(self, h, b, slots, n):
        for s in _mem(slots, n * 4, np.int32):
            self._bank(b)["rows"].pop(int(s), None)
        return 0

    def reid_bank_count(self, b, slot, out):
        out._obj.value = len(self._bank(b)["rows"].get(int(slot), []))
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
                    cost[i] = onn.nn_cosine_distance(np.stack(self._bank(bank)["rows"][int(sl[i])]), emb)
            mdist = max_dist.value if hasattr(max_dist, "value") else max_dist
            if mdist >= 0:
                cost[cost > mdist] = mdist + 1e-5
        if _v(tb) and t and m:
            iou = matching.diou_cost(_mem(tb, t * 32, np.float64).reshape(t, 4), _mem(db, m * 32, np.float64).reshape(m, 4))
        self.out[slot] = (emb.copy(), cost, iou)
        return 0

    def reid_frame_cost_groups(self, h, slot, groups, banks, t_counts, m_counts, slots, metric, max_dist, tb, db, want_emb):
        """bank.hip reid_frame_cost_groups step by step: group g's tracks against ITS detections, blocks concatenated."""
        fr = self.frame[slot]
        tc, mc = _mem(t_counts, groups * 4, np.int32), _mem(m_counts, groups * 4, np.int32)
        if int(mc.sum()) != fr["m"]:
            return -1
        t_all = int(tc.sum())
        sl = _mem(slots, t_all * 4, np.int32) if _v(slots) else None
        tboxes = _mem(tb, t_all * 32, np.float64).reshape(t_all, 4) if _v(tb) else None
        dboxes = _mem(db, fr["m"] * 32, np.float64).reshape(fr["m"], 4) if _v(db) else None
        mdist = max_dist.value if hasattr(max_dist, "value") else max_dist
        costs, ious, t_off, m_off = [], [], 0, 0
        for g in range(groups):
            t, m = int(tc[g]), int(mc[g])
            emb = fr["emb"][m_off:m_off + m]
            if t and m:
                if sl is not None and banks is not None:
                    bank = self._bank(banks[g])
                    c = np.empty((t, m), np.float32)
                    with np.errstate(invalid="ignore", divide="ignore"):
                        for i in range(t):
                            c[i] = onn.nn_cosine_distance(np.stack(bank["rows"][int(sl[t_off + i])]), emb)
                    if mdist >= 0:
                        c[c > mdist] = mdist + 1e-5
                    costs.append(c.reshape(-1))
                if tboxes is not None and dboxes is not None:
                    ious.append(matching.diou_cost(tboxes[t_off:t_off + t], dboxes[m_off:m_off + m]).reshape(-1))
            t_off += t
            m_off += m
        self.out[slot] = (fr["emb"][:fr["m"]].copy(), np.concatenate(costs) if costs else None, np.concatenate(ious) if ious else None)
        return 0

    def reid_frame_match_stream(self, h, on):
        return 0      # (the stand-in runs every stage when it is called: nothing to order)

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

    # ---- what bench.py's embed workload touches besides the above (tests/test_bench_launch.py)
    def reid_last_error(self):
        return b"stand-in library"

    def reid_ctx_create(self, device, out):
        out._obj.value = 1
        return 0

    def reid_ctx_destroy(self, h):
        return 0

    def _ok(self, *a):
        return 0

    reid_ctx_set_chunk = reid_ctx_set_precision = reid_profile_enable = reid_profile_reset = reid_timer_start = _ok
    reid_device_sync = reid_ctx_clear_fault = reid_ctx_set_stream = reid_ctx_set_null_stream = _ok

    def reid_seres18_load(self, h, blob, n, manifest):
        self.loads = getattr(self, "loads", 0) + 1
        hang_at = int(os.environ.get("REID_STANDIN_HANG_AT_LOAD", "0"))
        if hang_at and self.loads >= hang_at and self.rank == int(os.environ.get("REID_STANDIN_HANG_RANK", "0")):
            time.sleep(3600)                 # a rank that never comes back: what the bench's watchdog is for
        return 0

    def reid_seres18_dims(self, h, d, nc):
        d._obj.value, nc._obj.value = 512, 751
        return 0

    def reid_timer_stop(self, h, ms):
        ms._obj.value = 1.0
        return 0

    def reid_profile_get(self, h, kind, ms, n, fl, by):
        ms._obj.value, n._obj.value, fl._obj.value, by._obj.value = 1.0, 1, 1e9, 1e6
        return 0

    def reid_comm_unique_id(self, buf):
        return 0

    def reid_allreduce_f64(self, h, inout, count, op):
        if self.world == 1:
            return 0
        t = torch.tensor([inout[i] for i in range(count)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
        for i in range(count):
            inout[i] = float(t[i])
        return 0

