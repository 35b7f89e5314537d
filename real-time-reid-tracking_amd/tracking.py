"""Per-frame driver of the appearance path of one camera stream.

What it stands in for: `DeepSort.update` ([external] deep_sort.py, built by `build_tracker` from
modification_deepsort/deep_sort.yaml) calls, every frame, `Extractor.__call__` on the detections' crops
(modification_deepsort/feature_extractor.py:48-53), then `Tracker.update`, whose matching cascade asks
`NearestNeighborDistanceMetric.distance` and `iou_matching.iou_cost` for cost matrices and finally feeds the matched
features back with `partial_fit`.  Each of those is a blocking host call in the reference.  Here the four stages of
csrc/bank.hip (`reid_frame_submit / _cost / _fetch / _update`) are queued on the camera's HIP stream so that the device
embeds frame f+1 while the host still assigns frame f; the Kalman filter and the Hungarian assignment stay with the caller
(`assign`), they are not part of this path.

One `CameraStream` = one engine context (own stream, workspaces, feature bank).  A tracking frame leaves most CUs idle in
most of its launches, so several camera streams driven from several host threads overlap on one GPU.
"""
import numpy as np

from .engine import Engine, get_engine
from .nn_matching import NearestNeighborDistanceMetric


def _two_streams(stream, on):
    """`match_stream` of the stream classes: the cost and update stages of the context's frame pipeline (and every other access to its
    banks) on a second, high-priority HIP stream, ordered against the forwards by events (`reid_frame_match_stream`).  The forward of
    frame f + 1 then follows the forward of frame f directly instead of waiting behind cost(f), and update(f) / cost(f + 1) run in
    between the next forward's launches - which leave most of the chip idle anyway: 1.09 -> 1.12 k frames/s for the strict stream on
    the same box, and what makes a look-ahead group's chain run beside the next group's forward.  Results are bit-identical.
    For ONE stream object per GPU.  K `CameraStream` contexts on one device should pass `match_stream=False` (or better: be one
    `MultiCameraStream`): 3K streams on the device's few hardware queues make the cross-stream waits stall on queue switches - four
    contexts in four threads measured 1.35 k frames/s in total on one stream each, 1.15 k with match streams (two contexts: 1.07 / 1.10 k)."""
    stream.match_stream = bool(on)
    stream.eng.frame_match_stream(stream.match_stream)


def _one_stream_again(stream):
    if stream.match_stream and not stream._own:
        stream.eng.frame_match_stream(False)      # a shared context goes back to one stream


class CameraStream:
    def __init__(self, weights_blob, manifest, precision=0, max_dist=0.15, budget=100, metric="cosine", device=0, own_context=True,
                 max_tracks=4096, match_stream=True):
        self._own = bool(own_context)
        self.eng = Engine(device) if own_context else get_engine(device)
        self.eng.load_seres18(weights_blob, manifest)
        self.eng.set_precision(precision)
        _two_streams(self, match_stream)
        self.max_dist = max_dist
        self.metric = NearestNeighborDistanceMetric(metric, max_dist, budget, max_tracks=max_tracks, engine=self.eng)
        self._frame = 0

    def submit(self, crops):
        """Queue upload + embedding of the FIRST frame's crops (returns at once); later frames ride on `step(next_crops=...)`."""
        self.eng.frame_submit(self._frame & 1, crops)

    def step(self, targets, track_boxes, det_boxes, next_crops=None):
        """Costs of the submitted frame against the confirmed tracks `targets` (tlwh boxes for the DIoU cost), with the next
        frame's crops submitted in between so that the device works under the host's assignment.  Returns
        (features[m,512], appearance_cost[t,m] gated at max_dist, iou_cost[t,m] | None)."""
        slot = self._frame & 1
        self.metric.frame_distance_begin(slot, targets, self.max_dist, track_boxes, det_boxes)
        if next_crops is not None:
            self.eng.frame_submit(slot ^ 1, next_crops)
        return self.metric.frame_distance_end(slot)

    def commit(self, rows, targets, active_targets):
        """partial_fit: detection row rows[i] of the frame just stepped becomes a sample of track targets[i]; tracks missing
        from active_targets are forgotten.  Asynchronous.  Moves on to the next frame."""
        self.metric.frame_partial_fit(self._frame & 1, np.ascontiguousarray(rows, np.int32), targets, active_targets)
        self._frame += 1

    def close(self, destroy=False):
        """Drain the stream; destroy=True also frees the bank and, for an own context, the context."""
        self.eng.sync()
        _one_stream_again(self)
        if destroy:
            self.metric.close()
            if self._own:
                self.eng.close()


class MultiCameraStream:
    """K camera streams of ONE GPU batched into one pass per frame time.

    The reference runs one `track_yolov5.py` loop per video (modification_tracking/track_yolov5.py:178-253), each calling the
    extractor on its own ~30 crops; K `CameraStream`s on one device overlap poorly, because every launch of a frame is one
    wave of blocks that leaves most CUs idle for most of its life (four concurrent streams measured 1.28x one stream).  Here the
    K cameras' crops of a frame time are concatenated into ONE `reid_frame_submit` - a forward over ~30 K crops runs its
    convolutions as proper tiles - while everything that must stay per camera stays per camera: each has its own feature bank
    (`metrics[c]`), its tracks meet its own detections only (`reid_frame_cost_groups`: K small cost blocks, not a (sum t) x
    (sum m) matrix) and `commit` feeds each bank from the camera's slice of the slot.  Same call order as `CameraStream`,
    with one list entry per camera everywhere; one host thread, one wait per frame time."""

    def __init__(self, weights_blob, manifest, cameras, precision=0, max_dist=0.15, budget=100, metric="cosine", device=0,
                 own_context=True, max_tracks=4096, match_stream=True):
        self._own = bool(own_context)
        self.eng = Engine(device) if own_context else get_engine(device)
        self.eng.load_seres18(weights_blob, manifest)
        self.eng.set_precision(precision)
        _two_streams(self, match_stream)
        self.max_dist = max_dist
        self.cameras = int(cameras)
        self.metrics = [NearestNeighborDistanceMetric(metric, max_dist, budget, max_tracks=max_tracks, engine=self.eng)
                        for _ in range(self.cameras)]
        self._frame = 0
        self._m = {}                       # slot -> detections per camera of the submitted frame

    def _submit(self, slot, crops_per_camera):
        if len(crops_per_camera) != self.cameras:
            raise ValueError("expected %d crop lists, got %d" % (self.cameras, len(crops_per_camera)))
        self._m[slot] = [len(c) for c in crops_per_camera]
        self.eng.frame_submit(slot, [c for cam in crops_per_camera for c in cam])

    def submit(self, crops_per_camera):
        """Queue upload + embedding of the FIRST frame time's crops (one list per camera); later ones ride on `step`."""
        self._submit(self._frame & 1, crops_per_camera)

    def step(self, targets, track_boxes, det_boxes, next_crops=None):
        """Per camera c: costs of its submitted detections against its confirmed tracks ``targets[c]`` (tlwh boxes for the DIoU
        cost; ``track_boxes`` / ``det_boxes`` None = no DIoU), the next frame time's crops submitted in between.  Returns one
        (features[m_c,512], appearance_cost[t_c,m_c] gated at max_dist, iou_cost[t_c,m_c] | None) per camera."""
        slot = self._frame & 1
        ms = self._m[slot]
        groups = []
        for c, met in enumerate(self.metrics):
            met._ensure(512)
            tg = list(targets[c])
            groups.append((met._bank, met._slots_for(tg, False) if tg else np.empty(0, np.int32),
                           None if track_boxes is None else track_boxes[c], None if det_boxes is None else det_boxes[c], ms[c]))
        self.eng.frame_cost_groups(slot, groups, self.metrics[0]._metric, self.max_dist)
        if next_crops is not None:
            self._submit(slot ^ 1, next_crops)
        emb, costs, ious = self.eng.frame_fetch_groups(slot)
        out, off = [], 0
        for c in range(self.cameras):
            cost = costs[c] if costs[c] is not None else np.zeros((len(groups[c][1]), ms[c]), np.float32)
            out.append((emb[off:off + ms[c]], cost.astype(np.float64), ious[c]))
            off += ms[c]
        return out

    def commit(self, rows, targets, active_targets):
        """Per camera c: detection row rows[c][i] of ITS detections becomes a sample of its track targets[c][i]; its tracks missing
        from active_targets[c] are forgotten.  Asynchronous.  Moves on to the next frame time."""
        slot = self._frame & 1
        off = 0
        for c, met in enumerate(self.metrics):
            met.frame_partial_fit(slot, np.asarray(rows[c], np.int32) + off, targets[c], active_targets[c])
            off += self._m[slot][c]
        self._frame += 1

    def close(self, destroy=False):
        self.eng.sync()
        _one_stream_again(self)
        if destroy:
            for met in self.metrics:
                met.close()
            if self._own:
                self.eng.close()


class LookaheadCameraStream:
    """ONE camera stream whose detections are known F frames ahead - a video file or a detection dump (BASELINE configs[3] is one;
    `track_yolov5.py:178-253` reads its frames from a file and could run its detector a frame ahead): the crops of F consecutive
    frames go through the network as ONE pass, everything that depends on the tracker's state stays per frame and in order -
    frame j's costs are computed against the bank as frame j - 1's `commit` left it (`reid_frame_cost_groups` with one group per
    frame, only frame j's carrying tracks), so features, costs and bank contents are those of the frame-by-frame stream up to the
    pass-size effect on a crop's embedding (fp32 summation order).  The price is latency: a frame's features exist once the F-th
    frame of its group has been detected.  `CameraStream` (F = 1) remains the strict real-time form; this is the throughput form.

        s.submit_group([crops_f, crops_f+1, ...])                      # F lists of crops
        for j in range(F):
            feats, cost, iou = s.step(j, targets, track_boxes, det_boxes_of_frame_j, next_group=... if j == s.handover else None)
            s.commit(j, rows, targets, active_targets)

    `match_stream` (default): the cost and update stages run on a stream of their own (`reid_frame_match_stream`), the next group is
    handed over at the group's FIRST frame (`handover` = 0) and its forward runs on the compute stream beside the whole
    cost -> assign -> update chain of the current group: the device goes from one group's forward straight into the next one's
    (4 frames of ~30 crops per pass: 2.03 k frames/s against 1.51 k without, 2 frames: 1.58 k against 1.30 k).  Without it
    (`handover` = the group's last frame) everything is on one stream, as in `CameraStream`.  Results are identical either way.
    """

    def __init__(self, weights_blob, manifest, frames_per_pass=2, precision=0, max_dist=0.15, budget=100, metric="cosine", device=0,
                 own_context=True, max_tracks=4096, match_stream=True):
        self._own = bool(own_context)
        self.eng = Engine(device) if own_context else get_engine(device)
        self.eng.load_seres18(weights_blob, manifest)
        self.eng.set_precision(precision)
        _two_streams(self, match_stream)
        self.max_dist = max_dist
        self.frames_per_pass = int(frames_per_pass)
        self.metric = NearestNeighborDistanceMetric(metric, max_dist, budget, max_tracks=max_tracks, engine=self.eng)
        self._group = 0
        self._m = {}

    def _submit(self, slot, group):
        if not 1 <= len(group) <= self.frames_per_pass:
            raise ValueError("a group holds 1..%d frames, got %d" % (self.frames_per_pass, len(group)))
        self._m[slot] = [len(c) for c in group]
        self.eng.frame_submit(slot, [c for fr in group for c in fr])

    def submit_group(self, group):
        """Queue upload + embedding of the FIRST group of frames (a list of crop lists); later groups ride on the step of frame
        `handover` of the current group (`next_group=`).  On one stream that is the group's LAST frame: the stream then holds
        cost(last) | forward(next group) | update(last) and the device embeds the next group under the host's assignment of that
        frame only - handed over earlier, the forward would sit in front of the remaining frames' cost stages.  With the match
        stream nothing sits in front of anything and the hand-over is the first frame."""
        self._submit(self._group & 1, group)

    @property
    def handover(self):
        """The frame of the current group whose `step` should carry the next group."""
        return 0 if self.match_stream else len(self._m[self._group & 1]) - 1

    def step(self, j, targets, track_boxes, det_boxes, next_group=None):
        """Frame j of the submitted group: (features[m_j,512], appearance_cost[t,m_j] gated at max_dist, iou_cost[t,m_j] | None) against
        the confirmed tracks as the previous frame's commit left the bank."""
        slot = self._group & 1
        ms = self._m[slot]
        self.metric._ensure(512)
        tg = list(targets)
        slots = self.metric._slots_for(tg, False) if tg else np.empty(0, np.int32)
        boxes = track_boxes is not None and det_boxes is not None
        groups = []
        for f, m in enumerate(ms):
            mine = f == j
            groups.append((self.metric._bank, slots if mine else np.empty(0, np.int32),
                           (track_boxes if mine else np.empty((0, 4))) if boxes else None,
                           (det_boxes if mine else np.zeros((m, 4))) if boxes else None, m))
        self.eng.frame_cost_groups(slot, groups, self.metric._metric, self.max_dist)
        if next_group is not None:
            self._submit(slot ^ 1, next_group)
        emb, costs, ious = self.eng.frame_fetch_groups(slot)
        off = sum(ms[:j])
        cost = costs[j] if costs[j] is not None else np.zeros((len(tg), ms[j]), np.float32)
        return emb[off:off + ms[j]], cost.astype(np.float64), ious[j]

    def commit(self, j, rows, targets, active_targets):
        """partial_fit from frame j's rows of the group; after the group's last frame the stream moves on to the next group."""
        slot = self._group & 1
        self.metric.frame_partial_fit(slot, np.asarray(rows, np.int32) + sum(self._m[slot][:j]), targets, active_targets)
        if j == len(self._m[slot]) - 1:
            self._group += 1

    def close(self, destroy=False):
        self.eng.sync()
        _one_stream_again(self)
        if destroy:
            self.metric.close()
            if self._own:
                self.eng.close()


class ShardedCameraStream:
    """One camera stream whose frames are dealt over the ranks of a multi-GPU job (BASELINE configs[3]):
    every rank embeds its round-robin share of a frame's crops (`parallel.round_robin`), ONE device-side all-gather per frame
    (`reid_frame_gather`: equal blocks of ceil(n / world) rows, zero padding rows) puts the frame's features into the frame
    slot on EVERY rank, and every rank then computes the full cost matrices and keeps its own copy of the feature bank up to
    date - as DeepSORT would on every rank.  Same call order as `CameraStream` (submit, then per frame step + commit); all
    arguments and results are in DETECTION order, the mapping to rows of the gathered slot (`parallel.frame_rows`) stays inside.
    `engine` carries the rank's communicator (`RcclComm`); with world 1 this is `CameraStream` on an existing engine."""

    def __init__(self, engine, comm, max_dist=0.15, budget=100, metric="cosine", max_tracks=4096, match_stream=None):
        self.eng, self.rank, self.world = engine, int(comm.rank), int(comm.world)
        self._own = False
        # match_stream=None: on where no collective runs (one rank, no communicator).  With a communicator the all-gather stays on the
        # compute stream, behind the forward, and the slot's readers wait for it (reid_frame_gather re-records the slot's event) - correct
        # (loop-back ranks and a real 1-rank RCCL communicator, tests/test_gpu_parity.py) but SLOW beside librccl's own streams: with a real
        # 1-rank communicator the stream fell from 1 120 to 460 frames/s (bench.py --workload tracking, REID_BENCH_COMM1=1,
        # --match-stream 1 against 0) - the cross-stream waits stall on hardware-queue switches, as with K contexts.  Opt-in there.
        if match_stream is None:
            match_stream = self.world == 1 and not getattr(comm, "active", False)
        _two_streams(self, match_stream)
        self.max_dist = max_dist
        self.metric = NearestNeighborDistanceMetric(metric, max_dist, budget, max_tracks=max_tracks, engine=engine)
        self._frame = 0
        self._rows = None

    def share(self, crops):
        """This rank's crops of a frame: detections rank, rank + world, ..."""
        from .parallel import round_robin
        return [crops[int(i)] for i in round_robin(len(crops), self.world, self.rank)]

    def submit(self, crops):
        """Queue upload + embedding of this rank's share of the FIRST frame; later frames ride on `step(next_crops=...)`.
        ``crops``: the whole frame's crops, identical on every rank."""
        self.eng.frame_submit(self._frame & 1, self.share(crops))

    def step(self, n, targets, track_boxes=None, det_boxes=None, next_crops=None):
        """Frame of ``n`` detections (its crops were submitted before): gather, costs against `targets`, the next frame's share
        submitted in between.  Returns (features[n,512], appearance_cost[t,n], iou_cost[t,n] | None) in detection order."""
        from .parallel import frame_rows
        slot = self._frame & 1
        rows, per = frame_rows(n, self.world)
        self.eng.frame_gather(slot, per, self.world)
        dets = None
        if track_boxes is not None and det_boxes is not None:
            # the slot holds world * per rows; padding rows get a unit box (their cost columns are dropped below)
            dets = np.tile(np.asarray([0.0, 0.0, 1.0, 1.0]), (self.world * per, 1))
            dets[rows] = np.asarray(det_boxes, np.float64).reshape(-1, 4)[:n]
        self.metric.frame_distance_begin(slot, targets, self.max_dist, track_boxes if dets is not None else None, dets)
        if next_crops is not None:
            self.eng.frame_submit(slot ^ 1, self.share(next_crops))
        feats, cost, icost = self.metric.frame_distance_end(slot)
        self._rows = rows
        return feats[rows], cost[:, rows], None if icost is None else icost[:, rows]

    def commit(self, dets, targets, active_targets):
        """partial_fit: detection dets[i] of the frame just stepped becomes a sample of track targets[i] (on every rank's bank)."""
        rows = self._rows[np.asarray(dets, np.int64)] if len(dets) else np.empty(0, np.int64)
        self.metric.frame_partial_fit(self._frame & 1, rows.astype(np.int32), targets, active_targets)
        self._frame += 1

    def close(self, destroy=False):
        self.eng.sync()
        _one_stream_again(self)
        if destroy:
            self.metric.close()
