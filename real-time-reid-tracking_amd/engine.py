"""Thin numpy-facing wrapper over one libreid_hip context (one per process and device).

Nothing here computes: every method marshals numpy arrays (or raw device
pointers) into the C ABI and returns what the HIP kernels produced.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import check

_ENGINES = {}

IMG_H, IMG_W = 256, 128   # Extractor.size = (128, 256) as (W, H), feature_extractor.py:24


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Engine:
    def __init__(self, device=0):
        self.lib = _ffi.lib()
        h = C.c_void_p()
        check(self.lib.reid_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        self.embed_dim = 512
        self.num_class = 0

    def close(self):
        if self.h:
            self.lib.reid_ctx_sync(self.h)
            for bank in list(getattr(self, "_banks", [])):    # a bank belongs to its context: destroyed before it (no device leak
                bank.close()                                  # when the engine goes first)
            for p in getattr(self, "_pinned", []):
                self.lib.reid_host_free(self.h, p)
            self._pinned = []
            self.lib.reid_ctx_destroy(self.h)
            self.h = None

    def register_bank(self, metric):
        """Feature banks created on this engine (nn_matching.NearestNeighborDistanceMetric): `close` frees them first."""
        self.__dict__.setdefault("_banks", []).append(metric)

    def unregister_bank(self, metric):
        banks = self.__dict__.get("_banks", [])
        if metric in banks:
            banks.remove(metric)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- runtime
    def sync(self):
        check(self.lib.reid_ctx_sync(self.h))

    def device_sync(self):
        """hipDeviceSynchronize on this engine's device (every stream); raises if the context's fault word is set."""
        check(self.lib.reid_device_sync(self.h))

    def clear_fault(self):
        """Resets the sticky fault word (an activation outside f16's range in mode 2 / a non-finite embedding)."""
        check(self.lib.reid_ctx_clear_fault(self.h))

    def precision_ok(self, arch, mode):
        """Can the loaded checkpoint of ``arch`` (0 = ResNet18-IBN-SE family, 1 = Swin) run in ``mode``?  (reid_ctx_precision_ok)"""
        return self.lib.reid_ctx_precision_ok(self.h, int(arch), int(mode)) == 0

    def fault_bits(self):
        """The sticky fault word as bits (1 range, 2 non-finite embedding, 4 split-K rendezvous), without synchronising."""
        b = C.c_int()
        check(self.lib.reid_ctx_fault_peek(self.h, C.byref(b)))
        return b.value

    def set_stream(self, hip_stream):
        """Run on another HIP stream (0 / None = the context's own).  Work already enqueued on the old stream shares the
        context's workspaces with what follows, so a switch drains the old stream first."""
        new = int(hip_stream or 0)
        if new != getattr(self, "_stream", 0):
            self.sync()
            if new == -1:      # the HIP null stream (handle 0, which set_stream reads as "own")
                check(self.lib.reid_ctx_set_null_stream(self.h))
            else:
                check(self.lib.reid_ctx_set_stream(self.h, C.c_void_p(new)))
            self._stream = new

    def use_torch_stream(self):
        """Enqueue on torch's current stream of this device (CUDA-tensor entry points of the backbones).  torch's default
        stream is the HIP null stream, handle 0."""
        import torch
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.set_stream(s if s else -1)

    def on_torch_stream(self):
        """Context manager: run on torch's current stream inside, back on the stream the engine had before outside - the
        process-wide engine is shared with the host-path callers (Extractor, frame pipeline), which must not inherit torch's
        stream (with the null stream the copy side-stream is switched off)."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev = getattr(self, "_stream", 0)
            self.use_torch_stream()
            try:
                yield self
            finally:
                self.set_stream(prev)        # drains the torch stream first (set_stream syncs on a switch)
        return scope()

    def set_chunk(self, n):
        check(self.lib.reid_ctx_set_chunk(self.h, int(n)))

    def set_precision(self, mode):
        """0 exact fp32 (the reference's arithmetic), 1 fp16 storage / fp32 accumulate, 2 "fp32-class": fp32 storage, the 3x3
        stride-1 convolutions as three f16 matrix-core products per multiply on hi/lo-split operands (fp32 accumulate)."""
        check(self.lib.reid_ctx_set_precision(self.h, int(mode)))
        self._precision = int(mode)

    @property
    def precision(self):
        """The arithmetic mode this context is in (the library's default is 0)."""
        return getattr(self, "_precision", 0)

    def set_side_index(self, index):
        """Camera (ResNet18-IBN-SE: SERes18_IBN.py:269-270) or view (Swin: swin_transformer.py:301-302) index of every image of
        the following embed call; ``None`` / empty clears."""
        idx = np.ascontiguousarray(np.asarray([] if index is None else index).reshape(-1), np.int32)
        check(self.lib.reid_ctx_set_side_index(self.h, _ptr(idx) if idx.size else None, int(idx.size)))

    def malloc(self, nbytes):
        p = C.c_void_p()
        check(self.lib.reid_malloc(self.h, int(nbytes), C.byref(p)))
        return p.value

    def free(self, dptr):
        check(self.lib.reid_free(self.h, C.c_void_p(dptr)))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        check(self.lib.reid_memcpy_h2d(self.h, C.c_void_p(dptr), _ptr(arr), arr.nbytes))

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        check(self.lib.reid_memcpy_d2h(self.h, _ptr(arr), C.c_void_p(dptr), arr.nbytes))
        return arr

    def timer_start(self):
        check(self.lib.reid_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        check(self.lib.reid_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile(self, on):
        check(self.lib.reid_profile_enable(self.h, int(bool(on))))

    def profile_reset(self):
        check(self.lib.reid_profile_reset(self.h))

    def profile_get(self, kind):
        ms, n, fl, by = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
        check(self.lib.reid_profile_get(self.h, int(kind), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        return {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}

    # ---- weights
    def load_seres18(self, blob, manifest):
        blob = _f32(blob)
        self._owner = None                       # the plugin objects (Extractor, backbones) mark what THEY loaded afterwards
        check(self.lib.reid_seres18_load(self.h, _ptr(blob), blob.size, manifest.encode()))
        d, nc = C.c_int(), C.c_int()
        check(self.lib.reid_seres18_dims(self.h, C.byref(d), C.byref(nc)))
        self.embed_dim, self.num_class = d.value, nc.value

    def load_swin(self, blob, manifest):
        blob = _f32(blob)
        self._swin_owner = None
        check(self.lib.reid_swin_load(self.h, _ptr(blob), blob.size, manifest.encode()))
        d, nc = C.c_int(), C.c_int()
        check(self.lib.reid_swin_dims(self.h, C.byref(d), C.byref(nc)))
        self.swin_dim, self.swin_num_class = d.value, nc.value

    def swin_embed_f32_nchw(self, x, logits=False):
        """float32[n,3,h,w] (h, w multiples of 224) -> float32[n,96] (and logits)."""
        x = _f32(x)
        if x.ndim != 4 or x.shape[1] != 3 or x.shape[2] % 224 or x.shape[3] % 224:
            raise ValueError("swin_embed_f32_nchw expects float32[n,3,224k,224m], got %s" % (x.shape,))
        n = x.shape[0]
        emb = np.empty((n, self.swin_dim), np.float32)
        lg = np.empty((n, self.swin_num_class), np.float32) if logits else None
        check(self.lib.reid_swin_embed_f32_nchw(self.h, _ptr(x), n, x.shape[2], x.shape[3], _ptr(emb), _ptr(lg)))
        return (emb, lg) if logits else emb

    def swin_embed_dev(self, d_x, n, h, w, d_emb, d_logits=None):
        check(self.lib.reid_swin_embed_f32_nchw_dev(self.h, C.c_void_p(d_x), int(n), int(h), int(w), C.c_void_p(d_emb),
                                                    C.c_void_p(d_logits or 0)))

    # ---- embedding
    def _outs(self, n, want_logits):
        emb = np.empty((n, self.embed_dim), np.float32)
        logits = np.empty((n, self.num_class), np.float32) if want_logits else None
        return emb, logits

    def embed_u8(self, crops, logits=False):
        """uint8[n,256,128,3] -> float32[n,512] (and logits[n,num_class])."""
        crops = np.ascontiguousarray(crops, dtype=np.uint8)
        if crops.ndim != 4 or crops.shape[1:] != (IMG_H, IMG_W, 3):
            raise ValueError("embed_u8 expects uint8[n,%d,%d,3], got %s" % (IMG_H, IMG_W, crops.shape))
        emb, lg = self._outs(crops.shape[0], logits)
        check(self.lib.reid_embed_u8(self.h, _ptr(crops), crops.shape[0], _ptr(emb), _ptr(lg)))
        return (emb, lg) if logits else emb

    def embed_ragged_u8(self, crops, logits=False):
        """list of uint8[h_i,w_i,3] -> float32[n,512]; resize + normalise run on the device.  Crops that already lie one after
        the other in ONE host buffer (views of a stacked array, slices of a pinned slab) are handed over in place - no packing copy;
        more crops than a pass holds go up pass by pass under the kernels (csrc/api.hip reid_embed_ragged_u8)."""
        n = len(crops)
        hw = np.empty((n, 2), np.int32)
        offs = np.empty(n, np.int64)
        total = 0
        flat = []
        base = None                              # address of crop 0 while every crop so far starts where the previous one ended
        for i, c in enumerate(crops):
            c = np.ascontiguousarray(c, dtype=np.uint8)
            if c.ndim != 3 or c.shape[2] != 3 or c.shape[0] < 1 or c.shape[1] < 1:
                raise ValueError("crop %d must be uint8[h,w,3], got %s" % (i, c.shape))
            hw[i] = c.shape[:2]
            offs[i] = total
            addr = c.__array_interface__["data"][0]
            if i == 0:
                base = addr
            elif base is not None and addr != base + total:
                base = None
            total += c.size
            flat.append(c)
        emb, lg = self._outs(n, logits)
        if base is not None and n:
            src = C.c_void_p(base)               # `flat` keeps the views (and so their buffer) alive over the call
        else:
            packed = np.concatenate([c.reshape(-1) for c in flat]) if flat else np.empty(0, np.uint8)
            src = _ptr(packed)
        check(self.lib.reid_embed_ragged_u8(self.h, src, _ptr(offs), _ptr(hw), n, _ptr(emb), _ptr(lg)))
        return (emb, lg) if logits else emb

    # ---- frame pipeline (csrc/bank.hip): submit (asynchronous) / cost (the frame's one synchronisation) / update (asynchronous)
    def pinned(self, nbytes):
        """uint8 numpy array over pinned host memory (freed with the engine)."""
        p = C.c_void_p()
        check(self.lib.reid_host_alloc(self.h, int(nbytes), C.byref(p)))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(int(nbytes),))

    def frame_submit(self, slot, crops):
        """Stage 1: pack the ragged uint8 crops into the slot's pinned slab, enqueue upload + resize + forward; returns at once."""
        n = len(crops)
        hw = np.empty((n, 2), np.int32)
        offs = np.empty(n, np.int64)
        total = 0
        for i, c in enumerate(crops):
            if c.ndim != 3 or c.shape[2] != 3 or c.shape[0] < 1 or c.shape[1] < 1 or c.dtype != np.uint8:
                raise ValueError("crop %d must be uint8[h,w,3], got %s %s" % (i, c.dtype, c.shape))
            hw[i] = c.shape[:2]
            offs[i] = total
            total += c.size
        slabs = self.__dict__.setdefault("_slabs", {})
        slab = slabs.get(slot)
        if slab is None or slab.size < total:
            self.sync()          # the old slab may still be the source of an upload
            if slab is not None:                 # superseded: give its pinned memory back now, not at close()
                old = slab.ctypes.data
                for p in list(self._pinned):
                    if p.value == old:
                        self.lib.reid_host_free(self.h, p)
                        self._pinned.remove(p)
            slab = slabs[slot] = self.pinned(max(2 * total, 1 << 22))
        for i, c in enumerate(crops):   # straight into pinned memory: the only host copy of the pixels
            slab[offs[i]: offs[i] + c.size].reshape(c.shape)[...] = c
        check(self.lib.reid_frame_submit(self.h, int(slot), _ptr(slab), _ptr(offs), _ptr(hw), n))
        self.__dict__.setdefault("_frame_n", {})[slot] = n
        return n

    def frame_gather(self, slot, per, world):
        """Multi-GPU frames: all-gather the ranks' embeddings of the submitted frame into the slot (equal blocks of `per` rows);
        the slot then holds world * per rows on every rank (`parallel.frame_rows` maps detections to them)."""
        if world > 1:
            rank, w = C.c_int(), C.c_int()
            check(self.lib.reid_comm_info(self.h, C.byref(rank), C.byref(w)))
            if w.value != world:
                raise RuntimeError("frame_gather over %d ranks needs the C-ABI communicator (reid_comm_init); it spans %d" % (world, w.value))
        check(self.lib.reid_frame_gather(self.h, int(slot), int(per)))
        if per > 0:
            self._frame_n[slot] = int(world) * int(per)

    def frame_cost(self, slot, bank=None, slots=None, metric=0, max_dist=-1.0, track_boxes=None, det_boxes=None, want_emb=True):
        """Stage 2 (asynchronous): enqueue the appearance cost / DIoU cost of the submitted frame; `frame_fetch` collects."""
        m = self._frame_n[slot]
        t = 0 if slots is None else len(slots)
        tb = db = None
        if track_boxes is not None and det_boxes is not None:
            tb = np.ascontiguousarray(track_boxes, np.float64).reshape(-1, 4)
            db = np.ascontiguousarray(det_boxes, np.float64).reshape(-1, 4)
            if db.shape[0] != m:
                raise ValueError("det_boxes has %d rows for %d submitted crops" % (db.shape[0], m))
            if slots is not None and tb.shape[0] != t:
                raise ValueError("track_boxes has %d rows for %d tracks" % (tb.shape[0], t))
            t = tb.shape[0]
        sl = None if slots is None else np.ascontiguousarray(slots, np.int32)
        check(self.lib.reid_frame_cost(self.h, int(slot), bank if sl is not None else None, _ptr(sl), t, int(metric),
                                       C.c_float(max_dist), _ptr(tb), _ptr(db), 1 if want_emb else 0))
        self.__dict__.setdefault("_frame_q", {})[slot] = (m, t, want_emb, bank is not None and sl is not None and t and m,
                                                          tb is not None and t and m)

    def frame_fetch(self, slot):
        """The frame's one wait: (emb[m,512] | None, cost[t,m] float32 | None, iou_cost[t,m] float64 | None)."""
        m, t, want_emb, has_cost, has_iou = self._frame_q.pop(slot)
        emb = np.empty((m, 512), np.float32) if want_emb else None
        cost = np.empty((t, m), np.float32) if has_cost else None
        iou = np.empty((t, m), np.float64) if has_iou else None
        check(self.lib.reid_frame_fetch(self.h, int(slot), _ptr(emb) if m else None, _ptr(cost), _ptr(iou)))
        return emb, cost, iou

    def frame_match_stream(self, on=True):
        """Cost / update stages of the frame pipeline (and every other access to this context's banks) on a stream of their own
        (reid_frame_match_stream): a look-ahead group's frame-by-frame chain then runs beside the next group's forward."""
        check(self.lib.reid_frame_match_stream(self.h, 1 if on else 0))

    def frame_cost_groups(self, slot, groups, metric=0, max_dist=-1.0, want_emb=True):
        """Stage 2 for K camera streams batched into one slot (reid_frame_cost_groups): ``groups`` = one (bank, slots, track_boxes,
        det_boxes, m) per camera, in the order their crops were submitted; camera g's tracks meet ITS m detections only."""
        k = len(groups)
        banks = (C.c_void_p * k)()
        tc, mc = np.zeros(k, np.int32), np.zeros(k, np.int32)
        sl, tb, db = [], [], []
        boxes = all(g[2] is not None and g[3] is not None for g in groups)
        for i, (bank, slots, tboxes, dboxes, m) in enumerate(groups):
            banks[i] = bank
            tc[i], mc[i] = (0 if slots is None else len(slots)), int(m)
            if slots is not None:
                sl.append(np.asarray(slots, np.int32))
            if boxes:
                t4 = np.asarray(tboxes, np.float64).reshape(-1, 4)
                d4 = np.asarray(dboxes, np.float64).reshape(-1, 4)
                if t4.shape[0] != tc[i] or d4.shape[0] != mc[i]:
                    raise ValueError("camera %d: %d / %d boxes for %d tracks / %d crops" % (i, t4.shape[0], d4.shape[0], tc[i], mc[i]))
                tb.append(t4)
                db.append(d4)
        if int(mc.sum()) != self._frame_n[slot]:
            raise ValueError("the groups hold %d detections, the slot %d" % (int(mc.sum()), self._frame_n[slot]))
        sl = np.ascontiguousarray(np.concatenate(sl), np.int32) if sl else None
        tb = np.ascontiguousarray(np.concatenate(tb)) if boxes else None
        db = np.ascontiguousarray(np.concatenate(db)) if boxes else None
        check(self.lib.reid_frame_cost_groups(self.h, int(slot), k, banks if sl is not None else None, _ptr(tc), _ptr(mc), _ptr(sl), int(metric),
                                              C.c_float(max_dist), _ptr(tb), _ptr(db), 1 if want_emb else 0))
        tm = int((tc.astype(np.int64) * mc).sum())
        self.__dict__.setdefault("_frame_q", {})[slot] = (self._frame_n[slot], tm, want_emb, sl is not None and tm > 0, boxes and tm > 0, tc, mc)

    def frame_fetch_groups(self, slot):
        """The one wait of a batched frame: (emb[m,512] | None, [cost_g[t_g,m_g] float32 | None], [iou_g float64 | None])."""
        m, tm, want_emb, has_cost, has_iou, tc, mc = self._frame_q.pop(slot)
        emb = np.empty((m, 512), np.float32) if want_emb else None
        cost = np.empty(tm, np.float32) if has_cost else None
        iou = np.empty(tm, np.float64) if has_iou else None
        check(self.lib.reid_frame_fetch(self.h, int(slot), _ptr(emb) if m else None, _ptr(cost), _ptr(iou)))
        offs = np.concatenate([[0], np.cumsum(tc.astype(np.int64) * mc)])
        cut = lambda a: [None if a is None else a[offs[g]:offs[g + 1]].reshape(int(tc[g]), int(mc[g])) for g in range(len(tc))]
        return emb, cut(cost), cut(iou)

    def frame_update(self, slot, bank, rows, slots):
        """Stage 3: partial_fit from the slot's device-resident embeddings (row rows[i] -> track slot slots[i]); asynchronous."""
        rows = np.ascontiguousarray(rows, np.int32)
        slots = np.ascontiguousarray(slots, np.int32)
        check(self.lib.reid_frame_update(self.h, int(slot), bank, _ptr(rows), _ptr(slots), len(rows)))

    def embed_frame_u8(self, frame, boxes_xyxy, logits=False):
        """uint8[H,W,3] frame + int boxes [n,4] (x1,y1,x2,y2; crop = frame[y1:y2, x1:x2]) -> float32[n,512]."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        if frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError("frame must be uint8[H,W,3], got %s" % (frame.shape,))
        boxes = np.ascontiguousarray(boxes_xyxy, dtype=np.int32).reshape(-1, 4)
        n = boxes.shape[0]
        emb, lg = self._outs(n, logits)
        check(self.lib.reid_embed_frame_u8(self.h, _ptr(frame), frame.shape[0], frame.shape[1], _ptr(boxes), n, _ptr(emb),
                                           _ptr(lg)))
        return (emb, lg) if logits else emb

    def embed_f32_nchw(self, x, logits=False):
        x = _f32(x)
        if x.ndim != 4 or x.shape[1:] != (3, IMG_H, IMG_W):
            raise ValueError("embed_f32_nchw expects float32[n,3,%d,%d], got %s" % (IMG_H, IMG_W, x.shape))
        emb, lg = self._outs(x.shape[0], logits)
        check(self.lib.reid_embed_f32_nchw(self.h, _ptr(x), x.shape[0], _ptr(emb), _ptr(lg)))
        return (emb, lg) if logits else emb

    def embed_f32_nchw_dev(self, d_x, n, d_emb, d_logits=None):
        check(self.lib.reid_embed_f32_nchw_dev(self.h, C.c_void_p(d_x), int(n), C.c_void_p(d_emb), C.c_void_p(d_logits or 0)))

    def embed_u8_dev(self, d_crops, n, d_emb, d_logits=None):
        check(self.lib.reid_embed_u8_dev(self.h, C.c_void_p(d_crops), int(n), C.c_void_p(d_emb),
                                         C.c_void_p(d_logits or 0)))

    def debug_switch(self, name, value=None):
        """Experiment switch of this context through libreid_hip_debug.so (include/reid_hip_debug.h: reid_debug_set_switch); with
        ``value`` None returns the current value.  The product library itself takes no such switch from the environment."""
        dbg = _ffi.debug_lib()
        if value is None:
            v = C.c_longlong()
            check(dbg.reid_debug_get_switch(self.h, name.encode(), C.byref(v)))
            return v.value
        check(dbg.reid_debug_set_switch(self.h, name.encode(), C.c_longlong(int(value))))

    def debug_mfma_bare(self, shape, zero=False, iters=20000):
        """TFLOP/s of a registers-only f16 MFMA loop on this device (libreid_hip_debug.so, microbench.hip): shape 32 = 32x32x16,
        16 = 16x16x32; random operands unless ``zero``."""
        tf = C.c_float()
        check(_ffi.debug_lib().reid_debug_mfma_bare(self.h, int(shape), int(bool(zero)), int(iters), C.byref(tf)))
        return tf.value

    def debug_switches_from_env(self):
        """A/B tools: REID_DEBUG_SWITCHES="name=value,name=value" -> debug_switch calls (read by the TOOL, in Python)."""
        import os
        spec = os.environ.get("REID_DEBUG_SWITCHES", "")
        for item in filter(None, (t.strip() for t in spec.split(","))):
            name, _, val = item.partition("=")
            self.debug_switch(name.strip(), int(val))
        return spec

    def debug_keep(self, on):
        check(self.lib.reid_ctx_set_debug_keep(self.h, int(on)))

    def debug_stage(self, stage, n):
        sizes = [524288, 131072, 131072, 131072, 65536, 65536, 32768, 32768, 65536, 65536, 512]
        out = np.empty(sizes[stage] * n, np.float32)
        cnt = C.c_size_t()
        check(self.lib.reid_debug_stage(self.h, int(stage), _ptr(out), out.size, C.byref(cnt)))
        assert cnt.value == out.size
        return out

    def debug_swin_stage(self, stage, n, h=224, w=224):
        """Stage activations of the last Swin pass as NHWC arrays (0 sfe, 1..4 stage outputs, 5 GeM output [n,96])."""
        if stage == 5:
            shape = (n, 96)
        else:
            s = max(stage - 1, 0)
            shape = (n, (h // 4) >> s, (w // 4) >> s, 96 << s)
        out = np.empty(shape, np.float32)
        cnt = C.c_size_t()
        check(self.lib.reid_debug_swin_stage(self.h, int(stage), _ptr(out), out.size, C.byref(cnt)))
        assert cnt.value == out.size, (cnt.value, out.size)
        return out

    # ---- matching
    def distmat(self, x, y, metric=_ffi.METRIC_L2):
        x, y = _f32(x), _f32(y)
        if x.ndim != 2 or y.ndim != 2 or x.shape[1] != y.shape[1]:
            raise ValueError("distmat expects x[m,d], y[n,d]")
        out = np.empty((x.shape[0], y.shape[0]), np.float32)
        check(self.lib.reid_distmat(self.h, _ptr(x), x.shape[0], _ptr(y), y.shape[0], x.shape[1], int(metric), _ptr(out)))
        return out

    def distmat_dev(self, d_x, m, d_y, n, d, metric, d_out):
        check(self.lib.reid_distmat_dev(self.h, C.c_void_p(d_x), int(m), C.c_void_p(d_y), int(n), int(d), int(metric),
                                        C.c_void_p(d_out)))

    def argmin_rows(self, x, y, metric=_ffi.METRIC_L2):
        x, y = _f32(x), _f32(y)
        idx = np.empty(x.shape[0], np.int32)
        val = np.empty(x.shape[0], np.float32)
        check(self.lib.reid_argmin_rows(self.h, _ptr(x), x.shape[0], _ptr(y), y.shape[0], x.shape[1], int(metric),
                                        _ptr(idx), _ptr(val)))
        return idx, val

    def argmin_rows_dev(self, d_x, m, d_y, n, d, metric, d_idx, d_val=None):
        check(self.lib.reid_argmin_rows_dev(self.h, C.c_void_p(d_x), int(m), C.c_void_p(d_y), int(n), int(d), int(metric),
                                            C.c_void_p(d_idx), C.c_void_p(d_val or 0)))

    def knn(self, xq, xb, k):
        xq, xb = _f32(xq), _f32(xb)
        D = np.empty((xq.shape[0], k), np.float32)
        I = np.empty((xq.shape[0], k), np.int32)
        check(self.lib.reid_knn(self.h, _ptr(xq), xq.shape[0], _ptr(xb), xb.shape[0], xq.shape[1], int(k), _ptr(D), _ptr(I)))
        return D, I

    def knn_dev(self, d_xq, nq, d_xb, nb, d, k, d_D, d_I):
        check(self.lib.reid_knn_dev(self.h, C.c_void_p(d_xq), int(nq), C.c_void_p(d_xb), int(nb), int(d), int(k),
                                    C.c_void_p(d_D), C.c_void_p(d_I)))

    def descriptor_f32_nchw(self, x, flip_tta=True):
        """float32 [n,3,256,128] (normalised by the caller) -> float32 [n, 512 + num_class] retrieval descriptor."""
        x = _f32(x)
        if x.ndim != 4 or x.shape[1:] != (3, IMG_H, IMG_W):
            raise ValueError("descriptor_f32_nchw expects [n,3,%d,%d], got %s" % (IMG_H, IMG_W, x.shape))
        de, nc = self.embed_dim, self.num_class
        out = np.empty((x.shape[0], de + nc), np.float32)
        check(self.lib.reid_descriptor_f32_nchw(self.h, _ptr(x), x.shape[0], int(bool(flip_tta)), _ptr(out)))
        return out

    def descriptor_dev(self, d_x, n, flip_tta, d_out):
        check(self.lib.reid_descriptor_f32_nchw_dev(self.h, C.c_void_p(d_x), int(n), int(bool(flip_tta)), C.c_void_p(d_out)))

    def cam_debias_dev(self, d_x, cams, n, d, la=0.05, iters=0):
        cams = np.ascontiguousarray(cams, dtype=np.int32).reshape(-1)
        if cams.shape[0] != n:
            raise ValueError("cam_debias_dev: %d camera ids for %d rows" % (cams.shape[0], n))
        check(self.lib.reid_cam_debias_dev(self.h, C.c_void_p(d_x), _ptr(cams), int(n), int(d), C.c_float(la), int(iters)))

    def smooth_tracklets_dev(self, d_x, seqs, valid, n, d, keep=0.1):
        seqs = np.ascontiguousarray(seqs, dtype=np.int32).reshape(-1)
        v = None if valid is None else np.ascontiguousarray(np.asarray(valid).reshape(-1) != 0, dtype=np.uint8)
        if seqs.shape[0] != n or (v is not None and v.shape[0] != n):
            raise ValueError("smooth_tracklets_dev: seqs / valid must have %d entries" % n)
        check(self.lib.reid_smooth_tracklets_dev(self.h, C.c_void_p(d_x), _ptr(seqs), _ptr(v), int(n), int(d), C.c_float(keep)))

    def rank_eval_dev(self, d_qf, ql, qc, nq, d_gf, gl, gc, ng, d):
        """Features in HBM, labels / cameras host int64 arrays -> (cmc_sum int32[ng], ap float64[nq], valid int32[nq])."""
        ql, qc, gl, gc = (np.ascontiguousarray(a, dtype=np.int64) for a in (ql, qc, gl, gc))
        cmc = np.empty(ng, np.int32)
        ap = np.empty(nq, np.float64)
        valid = np.empty(nq, np.int32)
        check(self.lib.reid_rank_eval_dev(self.h, C.c_void_p(d_qf), _ptr(ql), _ptr(qc), int(nq), C.c_void_p(d_gf), _ptr(gl),
                                          _ptr(gc), int(ng), int(d), _ptr(cmc), _ptr(ap), _ptr(valid)))
        return cmc, ap, valid

    def cam_debias(self, x, cams, la=0.05, iters=0):
        """diminish_camera_bias (reid/inference_utils.py:5-15) on the device; returns a new float32 [n, d] array."""
        x = np.array(_f32(x), copy=True)
        cams = np.ascontiguousarray(cams, dtype=np.int32).reshape(-1)
        if x.ndim != 2 or cams.shape[0] != x.shape[0]:
            raise ValueError("cam_debias expects x[n,d] and cams[n]")
        check(self.lib.reid_cam_debias(self.h, _ptr(x), _ptr(cams), x.shape[0], x.shape[1], C.c_float(la), int(iters)))
        return x

    def smooth_tracklets(self, x, seqs, valid=None, keep=0.1):
        """smooth_tracklets (reid/inference_utils.py:18-27) on the device; returns a new float32 [n, d] array."""
        x = np.array(_f32(x), copy=True)
        seqs = np.ascontiguousarray(seqs, dtype=np.int32).reshape(-1)
        if x.ndim != 2 or seqs.shape[0] != x.shape[0]:
            raise ValueError("smooth_tracklets expects x[n,d] and seqs[n]")
        v = None if valid is None else np.ascontiguousarray(np.asarray(valid).reshape(-1) != 0, dtype=np.uint8)
        check(self.lib.reid_smooth_tracklets(self.h, _ptr(x), _ptr(seqs), _ptr(v), x.shape[0], x.shape[1], C.c_float(keep)))
        return x

    def rerank_jaccard(self, x, k1=20, k2=6, rank=None):
        """float32 [n, n] k-reciprocal Jaccard distance (reid/faiss_utils.py:147-244); rank: optional int32 [n, k1]."""
        x = _f32(x)
        n, d = x.shape
        out = np.empty((n, n), np.float32)
        if rank is not None:
            rank = np.ascontiguousarray(rank, dtype=np.int32)
            if rank.shape != (n, int(k1)):
                raise ValueError(f"rank must be [{n}, {k1}], got {rank.shape}")
        check(self.lib.reid_rerank_jaccard(self.h, _ptr(x), n, d, int(k1), int(k2), _ptr(rank) if rank is not None else None,
                                           _ptr(out)))
        return out

    def rerank_jaccard_dev(self, d_x, n, d, k1, k2, d_out, d_rank=None):
        check(self.lib.reid_rerank_jaccard_dev(self.h, C.c_void_p(d_x), int(n), int(d), int(k1), int(k2),
                                               C.c_void_p(d_rank or 0), C.c_void_p(d_out)))

    def diou(self, bbox, candidates):
        b = np.ascontiguousarray(bbox, dtype=np.float64).reshape(4)
        c = np.ascontiguousarray(candidates, dtype=np.float64).reshape(-1, 4)
        out = np.empty(c.shape[0], np.float64)
        check(self.lib.reid_diou(self.h, _ptr(b), _ptr(c), c.shape[0], _ptr(out)))
        return out

    def diou_cost(self, tracks, dets):
        t = np.ascontiguousarray(tracks, dtype=np.float64).reshape(-1, 4)
        d = np.ascontiguousarray(dets, dtype=np.float64).reshape(-1, 4)
        out = np.empty((t.shape[0], d.shape[0]), np.float64)
        check(self.lib.reid_diou_cost(self.h, _ptr(t), t.shape[0], _ptr(d), d.shape[0], _ptr(out)))
        return out

    def rank_eval(self, qf, ql, qc, gf, gl, gc):
        qf, gf = _f32(qf), _f32(gf)
        ql, qc, gl, gc = (np.ascontiguousarray(a, dtype=np.int64) for a in (ql, qc, gl, gc))
        nq, ng = qf.shape[0], gf.shape[0]
        cmc = np.empty(ng, np.int32)
        ap = np.empty(nq, np.float64)
        valid = np.empty(nq, np.int32)
        check(self.lib.reid_rank_eval(self.h, _ptr(qf), _ptr(ql), _ptr(qc), nq, _ptr(gf), _ptr(gl), _ptr(gc), ng,
                                      qf.shape[1], _ptr(cmc), _ptr(ap), _ptr(valid)))
        return cmc, ap, valid

    # ---- single operators
    def conv2d_nhwc(self, x, w, stride=1, pad=0, scale=None, shift=None, residual=None, relu=False):
        x, w = _f32(x), _f32(w)
        n, h, ww, cin = x.shape
        cout, r, s, _ = w.shape
        ho, wo = (h + 2 * pad - r) // stride + 1, (ww + 2 * pad - s) // stride + 1
        out = np.empty((n, ho, wo, cout), np.float32)
        sc = _f32(scale) if scale is not None else None
        sh = _f32(shift) if shift is not None else None
        res = _f32(residual) if residual is not None else None
        check(self.lib.reid_conv2d_nhwc(self.h, _ptr(x), n, h, ww, cin, _ptr(w), cout, r, s, stride, pad, _ptr(sc), _ptr(sh),
                                        _ptr(res), int(bool(relu)), _ptr(out)))
        return out

    def gemm_nt(self, a, b, bias=None):
        a, b = _f32(a), _f32(b)
        out = np.empty((a.shape[0], b.shape[0]), np.float32)
        bi = _f32(bias) if bias is not None else None
        check(self.lib.reid_gemm_nt(self.h, _ptr(a), a.shape[0], _ptr(b), b.shape[0], a.shape[1], _ptr(bi), _ptr(out)))
        return out


def get_engine(device=0):
    """Process-wide engine per device (one context per (process, device))."""
    device = int(device)
    if device not in _ENGINES:
        _ENGINES[device] = Engine(device)
    return _ENGINES[device]
