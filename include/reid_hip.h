/*
 * reid_hip.h - C ABI of the MI355X (gfx950) re-ID embedding-and-matching engine.
 *
 * The reference (SuperbTUM/real-time-ReID-tracking) is pure Python and has no FFI
 * of its own; the entry points below are what a binding for its DeepSORT hot
 * path needs.  Each one cites the reference interface it
 * replaces.  INTEGRATION.md shows the ctypes stub a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   - every function returns 0 (REID_OK) or a negative reid_status; the message
 *     for the calling thread is available from reid_last_error().
 *   - "host" entry points take plain host pointers, stage through library-owned
 *     device buffers and return after the result is in the caller's memory.
 *   - "_dev" entry points take device pointers (hipMalloc / torch.Tensor.data_ptr()),
 *     enqueue on the context's stream and return without synchronising.
 *   - outputs are caller-allocated; the library never frees caller memory.
 *   - one context per (process, device); a context is not thread-safe (the
 *     reference's caller is single-threaded: feature_extractor.py:48-53).
 *   - layouts: crops are NHWC uint8 (the DeepSORT caller's HxWx3 arrays);
 *     float images are NCHW fp32 (torch convention of the plugin surface);
 *     matrices are row-major fp32.
 */
#ifndef REID_HIP_H
#define REID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct reid_ctx reid_ctx;

enum reid_status {
    REID_OK = 0,
    REID_ERR_ARG = -1,     /* bad argument (null pointer, bad size, unknown metric) */
    REID_ERR_HIP = -2,     /* a HIP runtime call failed */
    REID_ERR_STATE = -3,   /* e.g. embed before weights were loaded */
    REID_ERR_NOMEM = -4
};

/* distance definitions (metric argument) */
enum reid_metric {
    REID_METRIC_L2 = 0,        /* sqrt(clamp(|x|^2+|y|^2-2x.y, 1e-12))   reid/losses/utils.py:21-35 euclidean_dist */
    REID_METRIC_L2SQR = 1,     /* |x|^2+|y|^2-2x.y                        faiss METRIC_L2, reid/faiss_utils.py:56-118 */
    REID_METRIC_COS_HALF = 2,  /* (1 - x.y/(|x||y|))/2                    reid/losses/utils.py:12-18 cosine_dist */
    REID_METRIC_COS = 3,       /* 1 - x.y/(|x||y|)                        DeepSORT nn_matching cosine (MAX_DIST, deep_sort.yaml:3) */
    REID_METRIC_DOT = 4        /* x.y (similarity)                        reid/evaluate.py:58 score = gf @ q */
};

/* kernel classes for reid_profile_get() */
enum reid_kernel_kind {
    REID_K_CONV_GEMM = 0,   /* implicit-GEMM convolutions (the dominant kernel of the embed path) */
    REID_K_DIST_GEMM = 1,   /* N x M distance-matrix GEMM */
    REID_K_ELEMENTWISE = 2, /* norm / SE / pool / combine kernels */
    REID_K_SELECT = 3,      /* argmin / top-k / rank counting */
    REID_K_COUNT = 4
};

/* ---- runtime ---------------------------------------------------------------------------- */
const char* reid_last_error(void);
int reid_device_count(int* n);
/* one context per process and device; replaces `self.device` / `.to(device)` of feature_extractor.py:17,22 */
int reid_ctx_create(int device, reid_ctx** out);
int reid_ctx_destroy(reid_ctx* ctx);
/* run on an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL = the context's own stream */
int reid_ctx_set_stream(reid_ctx* ctx, void* hip_stream);
/* run on the HIP null (legacy default) stream - torch's default stream; its handle 0 means "own stream" to reid_ctx_set_stream */
int reid_ctx_set_null_stream(reid_ctx* ctx);
int reid_ctx_sync(reid_ctx* ctx);
/* hipDeviceSynchronize on the context's device: every stream, not only the context's (the timing bracket of bench.py; replaces
 * the torch.cuda.synchronize() a torch host would call around track_yolov5.py:178-253's loop) */
int reid_device_sync(reid_ctx* ctx);
/* The context's sticky fault word.  Kernels raise it when (a) in mode 2 an activation outside f16's range (or a NaN) reaches a
 * site that splits operands into [xh | xl'], or (b) in any mode a non-finite embedding leaves the neck - a checkpoint that
 * load_pretrained_weights (modification_tracking/reid_model_factory.py:158-210) accepted but this arithmetic cannot run.  While it
 * is set, every embed / frame entry point, reid_ctx_sync and reid_device_sync return REID_ERR_STATE (the synchronising embed
 * calls report a fault raised by their own work); reid_ctx_clear_fault drains the stream and resets it. */
int reid_ctx_clear_fault(reid_ctx* ctx);
/* crops per pass through the network, 1..4096; default 1024 (the measured-best size: a pass of 1024 crops fills every launch of the
 * network with whole rounds of tiles; activation working set = crops in the pass * 3.2 MB).  Swin passes are capped at 1024 images. */
int reid_ctx_set_chunk(reid_ctx* ctx, int crops_per_pass);
/* arithmetic of the convolution GEMMs:
 *   0 = exact fp32 (v_mfma_f32_32x32x2_f32) - the reference's arithmetic, the default;
 *   1 = fp16 storage / fp32 accumulate (activations and weights live in HBM as f16; north_star's 1e-3 cosine tolerance);
 *   2 = "fp32-class": fp32 storage, every convolution (and every Linear / convolution of the Swin trunk) as three
 *       v_mfma_f32_32x32x16_f16 per multiply on hi/lo-split operands (x = xh + xl, xh = f16(x), xl' = f16((x - xh) 2^11)) with fp32 accumulation - 22-bit operands, the
 *       dropped xl.wl term at 2^-22; held to mode 0's parity thresholds (tests/test_gpu_parity.py).  Operands must lie inside
 *       f16's range: |activation| < 65504, |weight| 2^11 < 65504 (any trained ResNet18-IBN-SE by a wide margin).  ENFORCED: mode 2
 *       is refused (REID_ERR_ARG naming the tensor) when the loaded checkpoint has a convolution / Linear weight outside the
 *       range, and so is loading such a checkpoint into a context already in mode 2; an activation outside the range raises the
 *       context's fault word (reid_ctx_clear_fault below). */
int reid_ctx_set_precision(reid_ctx* ctx, int mode);
/* The shared context of a process may hold the weights of BOTH backbones.  reid_ctx_set_precision(2) refuses when either loaded
 * checkpoint is outside the range; a host object that owns one of them asks here whether ITS checkpoint is the reason: arch 0 =
 * ResNet18-IBN-SE family, 1 = Swin; REID_OK when that checkpoint can run in `mode` (or none is loaded), REID_ERR_ARG + the
 * tensor's name otherwise.  reid_ctx_fault_peek: the sticky fault word without draining the stream or starting work (bit 0 range,
 * bit 1 non-finite embedding, bit 2 split-K rendezvous) - lets a caller tell a fault that predates its call from one it raised.
 * (No reference counterpart: the reference has one arithmetic, modification_deepsort/feature_extractor.py:15-29.) */
int reid_ctx_precision_ok(reid_ctx* ctx, int arch, int mode);
int reid_ctx_fault_peek(reid_ctx* ctx, int* bits);
/* optional side information of the NEXT embed call(s): one index per image, consumed in order by the passes that follow (n images
 * in total; n = 0 clears).  ResNet18-IBN-SE: the camera of every crop - SERse18_IBN.forward(x, cam) adds cam_factor *
 * cam_bias[cam] to the BNNeck output before the classifier (reid/backbones/SERes18_IBN.py:269-270); Swin: the view of every
 * image - SwinTransformer.forward(img, view_index) adds side_info_coeff * side_info_embedding[view] to the SFE output
 * (reid/backbones/swin_transformer.py:301-302).  The weight blob must carry the table (cam.bias + cam.factor / sfe.side +
 * sfe.side_coeff); an index outside it, or more images than indices, is REID_ERR_ARG. */
int reid_ctx_set_side_index(reid_ctx* ctx, const int32_t* index, int n);

/* device memory for hosts without torch */
int reid_malloc(reid_ctx* ctx, size_t bytes, void** dptr);
int reid_free(reid_ctx* ctx, void* dptr);
int reid_memcpy_h2d(reid_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int reid_memcpy_d2h(reid_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* pinned host memory: uploads from it are asynchronous DMAs (a copy from pageable memory blocks the caller until the
 * stream has drained) - where a tracker packs the crops it hands to reid_frame_submit */
int reid_host_alloc(reid_ctx* ctx, size_t bytes, void** out);
int reid_host_free(reid_ctx* ctx, void* p);

/* HIP-event timing on the context's stream (bench.py) */
int reid_timer_start(reid_ctx* ctx);
int reid_timer_stop(reid_ctx* ctx, float* elapsed_ms);      /* records, synchronises, returns elapsed */
/* per-kernel-class timing: when enabled every launch of the class is bracketed by HIP events */
int reid_profile_enable(reid_ctx* ctx, int on);
int reid_profile_reset(reid_ctx* ctx);
int reid_profile_get(reid_ctx* ctx, int kind, double* total_ms, long long* launches, double* flops, double* bytes);

/* ---- weights ----------------------------------------------------------------------------- */
/* Packed ResNet18-IBN-SE weights (reid_amd.weights.pack_seres18): one fp32 blob plus a text manifest of
 * "name offset count" lines.  Replaces torch.load + load_state_dict(strict=False), feature_extractor.py:18-19,
 * and load_pretrained_weights, modification_tracking/reid_model_factory.py:158-210. */
int reid_seres18_load(reid_ctx* ctx, const float* blob, size_t n_floats, const char* manifest);
int reid_seres18_dims(reid_ctx* ctx, int* embed_dim, int* num_class);

/* ---- embedding: Extractor.__call__, feature_extractor.py:48-53; SERse18_IBN.forward, SERes18_IBN.py:250-276 */
/* n crops already at 128x256: uint8[n][256][128][3] -> emb fp32[n][512] (BNNeck output), logits fp32[n][num_class] or NULL */
int reid_embed_u8(reid_ctx* ctx, const uint8_t* crops_nhwc, int n, float* emb, float* logits);
int reid_embed_u8_dev(reid_ctx* ctx, const uint8_t* d_crops_nhwc, int n, float* d_emb, float* d_logits);
/* crops of arbitrary size (feature_extractor.py:31-46 _preprocess: /255, bilinear resize to 128x256, (x-0.5)/0.5):
 * packed = concatenated HxWx3 uint8 images, offsets[i] = byte offset of crop i, hw[2i],hw[2i+1] = its height,width */
int reid_embed_ragged_u8(reid_ctx* ctx, const uint8_t* packed, const int64_t* offsets, const int32_t* hw, int n,
                         float* emb, float* logits);
/* crops as windows frame[y1:y2, x1:x2] of ONE uint8 HxWx3 frame (what DeepSort._get_features slices before calling the
 * Extractor): the frame is uploaded once and each window is resized on the device.  boxes_xyxy int32[n][4], inside the
 * frame and non-empty (an empty slice fails in the reference's cv2.resize as well). */
int reid_embed_frame_u8(reid_ctx* ctx, const uint8_t* frame_hwc, int fh, int fw, const int32_t* boxes_xyxy, int n,
                        float* emb, float* logits);
/* normalised float images, fp32[n][3][256][128] NCHW: the model(im_batch) call of the plugin surface
 * (modification_tracking/models/__init__.py:93-121 returned object) */
int reid_embed_f32_nchw(reid_ctx* ctx, const float* x, int n, float* emb, float* logits);
int reid_embed_f32_nchw_dev(reid_ctx* ctx, const float* d_x, int n, float* d_emb, float* d_logits);
/* intermediate activation of the last embed call, for stage-level parity tests:
 * stage 0 = stem conv+BN [n][128][64][64], 1 = maxpool [n][64][32][64], 2..9 = SE blocks 11..42 (NHWC), 10 = GeM [n][512].
 * Only valid when n <= chunk and after reid_ctx_set_debug_keep(ctx, 1).  Copies up to max_floats and returns the stage's element count in *count. */
int reid_ctx_set_debug_keep(reid_ctx* ctx, int on);   /* tests only: give every stage its own buffer; 1 also splits the
                                                        * fp16 stem into conv + pool kernels so that stage 0 exists,
                                                        * 2 keeps the production (fused) kernels: stage 0 is not written */
int reid_debug_stage(reid_ctx* ctx, int stage, float* out_host, size_t max_floats, size_t* count);
/* the same for the Swin backbone: stage 0 = ShadowFeatureExtraction output [n][H/4][W/4][96], 1..4 = the four stage outputs (NHWC),
 * 5 = GeM_1D output [n][96]; valid after a reid_swin_embed_* call that ran as one pass (n <= min(chunk, 1024)) */
int reid_debug_swin_stage(reid_ctx* ctx, int stage, float* out_host, size_t max_floats, size_t* count);

/* ---- Swin-T backbone (reference "v1": reid/backbones/swin_transformer.py:339-427, swin_t :508-513) -------------------
 * Packed weights from reid_amd.weights.pack_swin.  Input: normalised float images fp32[n][3][h][w] NCHW with h, w multiples
 * of 224 (the reference rejects 128x256); output: 96-d BatchNorm'd embedding (x_norm, :421) and logits. */
int reid_swin_load(reid_ctx* ctx, const float* blob, size_t n_floats, const char* manifest);
int reid_swin_dims(reid_ctx* ctx, int* embed_dim, int* num_class);
int reid_swin_embed_f32_nchw(reid_ctx* ctx, const float* x, int n, int h, int w, float* emb, float* logits);
int reid_swin_embed_f32_nchw_dev(reid_ctx* ctx, const float* d_x, int n, int h, int w, float* d_emb, float* d_logits);

/* ---- matching ----------------------------------------------------------------------------- */
/* out[m][n] = metric(x[m][d], y[n][d])        reid/losses/utils.py:12-35, reid/evaluate.py:58 */
int reid_distmat(reid_ctx* ctx, const float* x, int m, const float* y, int n, int d, int metric, float* out);
int reid_distmat_dev(reid_ctx* ctx, const float* d_x, int m, const float* d_y, int n, int d, int metric, float* d_out);
/* per-row minimum of the distance matrix; ties -> lowest column index.  From 2048 rows of y on the selection runs in the
 * epilogue of the distance GEMM (csrc/dist_select.hip) and the m x n matrix is never written; smaller problems go through a
 * scratch matrix of a few MB.  evaluate.py:58-63 (the top-1 of `score`). */
int reid_argmin_rows(reid_ctx* ctx, const float* x, int m, const float* y, int n, int d, int metric,
                     int32_t* idx, float* val);
int reid_argmin_rows_dev(reid_ctx* ctx, const float* d_x, int m, const float* d_y, int n, int d, int metric,
                         int32_t* d_idx, float* d_val);
/* brute-force squared-L2 k-NN: D fp32[nq][k] ascending, I int32[nq][k]; ties -> lowest index.  k <= 64 and nb >= 2048: fused
 * distance + selection (no nq x nb matrix), otherwise distance matrix tiles + a top-k pass.
 * search_raw_array_pytorch / IndexFlatL2.search, reid/faiss_utils.py:56-139 */
int reid_knn(reid_ctx* ctx, const float* xq, int nq, const float* xb, int nb, int d, int k, float* D, int32_t* I);
int reid_knn_dev(reid_ctx* ctx, const float* d_xq, int nq, const float* d_xb, int nb, int d, int k,
                 float* d_D, int32_t* d_I);
/* k-reciprocal Jaccard re-ranking, compute_jaccard_distance reid/faiss_utils.py:147-244: x fp32[n][d] L2-normalised rows,
 * rank int32[n][k1] neighbour lists or NULL (then reid_knn supplies them, self included), out fp32[n][n].
 * 1 <= k1 <= 64, k1 <= n, k2 >= 1 (k2 == 1 skips the local query expansion, as in the reference). */
int reid_rerank_jaccard(reid_ctx* ctx, const float* x, int n, int d, int k1, int k2, const int32_t* rank, float* out_nn);
int reid_rerank_jaccard_dev(reid_ctx* ctx, const float* d_x, int n, int d, int k1, int k2, const int32_t* d_rank,
                            float* d_out_nn);
/* DIoU of one tlwh box against m candidates, fp64, bit-exact with numpy   modification_deepsort/iou_matching.py:5-47 */
int reid_diou(reid_ctx* ctx, const double* box4, const double* cand_m4, int m, double* out_m);
/* cost[t][m] = 1 - DIoU(tracks[t], dets[m])   ([external] deep_sort iou_cost loop over iou()) */
int reid_diou_cost(reid_ctx* ctx, const double* tracks_t4, int t, const double* dets_m4, int m, double* out_tm);
/* ---- evaluation-script post-processing --------------------------------------------------- */
/* retrieval descriptor of reid/image_reid_inference.py:112-123,252-253: cat(normalize(emb), normalize(logits)) per image,
 * with flip_tta != 0 averaged with the horizontally mirrored image and renormalised.  x: fp32 [n][3][256][128] already
 * normalised by the caller's transform; out: fp32 [n][512 + num_class] (reid_seres18_dims). */
int reid_descriptor_f32_nchw(reid_ctx* ctx, const float* x, int n, int flip_tta, float* out);
int reid_descriptor_f32_nchw_dev(reid_ctx* ctx, const float* d_x, int n, int flip_tta, float* d_out);
/* diminish_camera_bias, reid/inference_utils.py:5-15, in place on x fp32 [n][d]: per camera id c (cams is a host int32[n],
 * ids >= 0) rows X_c <- normalize_rows((X_c - mean X_c) inverse(X_c^T X_c + n_c*la*I)^T).  The inverse is a Newton-Schulz
 * iteration of fp32 GEMMs that stops when the residual reaches the fp32 noise floor (iters = upper bound, <= 0: 40). */
int reid_cam_debias(reid_ctx* ctx, float* x, const int32_t* cams, int n, int d, float la, int iters);
int reid_cam_debias_dev(reid_ctx* ctx, float* d_x, const int32_t* cams, int n, int d, float la, int iters);
/* smooth_tracklets, reid/inference_utils.py:18-27, in place on x fp32 [n][d]: every row with valid[i] != 0 (valid NULL = all)
 * becomes keep * row + (1 - keep) * mean of the valid rows that share its sequence id (keep = 0.1 in the reference).
 * seqs / valid are host arrays. */
int reid_smooth_tracklets(reid_ctx* ctx, float* x, const int32_t* seqs, const uint8_t* valid, int n, int d, float keep);
int reid_smooth_tracklets_dev(reid_ctx* ctx, float* d_x, const int32_t* seqs, const uint8_t* valid, int n, int d, float keep);
/* ---- DeepSORT appearance metric with the feature bank on the device ---------------------------
 * [external] deep_sort/sort/nn_matching.py NearestNeighborDistanceMetric (the per-frame consumer of Extractor.__call__;
 * MAX_DIST / NN_BUDGET from modification_deepsort/deep_sort.yaml:3,9).  A bank holds, per track slot, the last `budget`
 * d-dimensional features in HBM; the caller maps track ids to slots 0..max_tracks-1. */
typedef struct reid_bank reid_bank;
int reid_bank_create(reid_ctx* ctx, int max_tracks, int budget, int d, reid_bank** out);
int reid_bank_destroy(reid_bank* bank);   /* before reid_ctx_destroy of its context */
/* partial_fit: row i of feats[n][d] is appended to track slots[i], in call order (oldest samples fall out of the ring) */
int reid_bank_update(reid_ctx* ctx, reid_bank* bank, const float* feats, const int32_t* slots, int n);
int reid_bank_update_dev(reid_ctx* ctx, reid_bank* bank, const float* d_feats, const int32_t* slots, int n);
/* forget tracks (targets missing from active_targets) so their slots can be reused */
int reid_bank_clear(reid_ctx* ctx, reid_bank* bank, const int32_t* slots, int n);
/* samples currently held for a slot (<= budget) */
int reid_bank_count(reid_bank* bank, int slot, int* out);
/* distance(): cost[t][m] = min over the samples of track slots[t] of metric(sample, dets[m]); metric REID_METRIC_COS
 * (_nn_cosine_distance) or REID_METRIC_L2SQR (_nn_euclidean_distance).  max_dist >= 0 also applies
 * linear_assignment.min_cost_matching's gate: cost > max_dist -> max_dist + 1e-5.  slots is a host array. */
int reid_bank_cost(reid_ctx* ctx, reid_bank* bank, const int32_t* slots, int t, const float* dets, int m, int metric,
                   float max_dist, float* out_tm);
int reid_bank_cost_dev(reid_ctx* ctx, reid_bank* bank, const int32_t* slots, int t, const float* d_dets, int m, int metric,
                       float max_dist, float* d_out_tm);
/* ---- one DeepSORT frame as asynchronous stages with a single wait ------------------------------
 * [external] deep_sort.py DeepSort.update: _get_features (Extractor.__call__, feature_extractor.py:48-53) -> Tracker.update
 * -> metric.distance + iou_cost -> metric.partial_fit.  `slot` (0 / 1) names one of two frame slots that own their crops,
 * embeddings and result staging: call cost(f), submit(f+1), fetch(f), <assign on the host>, update(f) and the device
 * embeds frame f+1 while the host assigns frame f.
 * submit (asynchronous): crops as for reid_embed_ragged_u8; `packed` must stay untouched until the slot's reid_frame_fetch
 *   returns (offsets / hw are copied).  m == 0 is allowed.
 * cost (asynchronous): appearance cost as reid_bank_cost over track slots[t] (bank NULL: skipped), DIoU cost as
 *   reid_diou_cost over tlwh boxes tracks_t4[t] / dets_m4[m] (NULL: skipped), embeddings when want_emb != 0.  The full
 *   t x m matrices serve every subset the matching cascade asks for.
 * fetch (waits for the slot's cost stage only): emb fp32[m][512], cost_tm fp32[t][m], iou_tm fp64[t][m]; NULL = not wanted.
 * update (asynchronous): partial_fit with row rows[i] of the slot's embeddings appended to track slots[i]. */
int reid_frame_submit(reid_ctx* ctx, int slot, const uint8_t* packed, const int64_t* offsets, const int32_t* hw, int m);
int reid_frame_cost(reid_ctx* ctx, int slot, reid_bank* bank, const int32_t* slots, int t, int metric, float max_dist,
                    const double* tracks_t4, const double* dets_m4, int want_emb);
int reid_frame_fetch(reid_ctx* ctx, int slot, float* emb, float* cost_tm, double* iou_tm);
/* K camera streams batched into ONE pass per frame time (track_yolov5.py:178-253 runs one such loop per video; a frame of ~30
 * crops leaves most of the chip idle, K cameras' frames in one forward do not): submit the cameras' crops one camera after the
 * other (m_counts[g] crops of camera g, sum = the slot's m), then this cost stage in place of reid_frame_cost - camera g's
 * t_counts[g] tracks (slots / tracks_t4: the groups' concatenations, on banks[g]) against ITS detections only.  reid_frame_fetch
 * then returns, in cost_tm / iou_tm, the groups' t_g x m_g blocks one after the other; reid_frame_update takes slot-wide rows. */
int reid_frame_cost_groups(reid_ctx* ctx, int slot, int groups, reid_bank* const* banks, const int32_t* t_counts,
                           const int32_t* m_counts, const int32_t* slots, int metric, float max_dist, const double* tracks_t4,
                           const double* dets_m4, int want_emb);
/* multi-GPU frames (every rank embeds its round-robin share of the frame's crops): between submit and
 * cost, gather the ranks' embeddings into the slot as equal blocks of per = ceil(n / world) rows - one ncclAllGather; the slot
 * then holds world * per rows (row r * per + i = detection r + i * world; rows past a rank's share are padding) on every rank.
 * Asynchronous; without a communicator (reid_comm_init) a no-op. */
int reid_frame_gather(reid_ctx* ctx, int slot, int per);
int reid_frame_update(reid_ctx* ctx, int slot, reid_bank* bank, const int32_t* rows, const int32_t* slots, int n);
/* Look-ahead streams (a video file / detection dump whose detections are known F frames ahead: F frames' crops are submitted as one
 * slot, their cost / update stages follow frame by frame): on != 0 moves the cost and update stages of THIS context's frame pipeline
 * to a stream of their own, ordered against the forwards by events, so that a group's serial cost -> assign -> update chain runs
 * beside the next group's forward.  While it is on, the context's banks must be driven through reid_frame_cost* / reid_frame_update
 * only.  Synchronises the context.  ([external] deep_sort.py DeepSort.update is strictly frame by frame: no counterpart.) */
int reid_frame_match_stream(reid_ctx* ctx, int on);
/* retrieval evaluation, reid/evaluate.py:33-105: for every query the ranks of its good gallery items among
 * non-junk items (descending similarity gf@q).  cmc_sum int32[ng] = sum over valid queries of the CMC step,
 * ap double[nq], valid int32[nq] (0 when the query has no good item). */
int reid_rank_eval(reid_ctx* ctx, const float* qf, const int64_t* ql, const int64_t* qc, int nq,
                   const float* gf, const int64_t* gl, const int64_t* gc, int ng, int d,
                   int32_t* cmc_sum, double* ap, int32_t* valid);
/* the same with the features already on the device (labels and results stay host arrays; synchronises) */
int reid_rank_eval_dev(reid_ctx* ctx, const float* d_qf, const int64_t* ql, const int64_t* qc, int nq,
                       const float* d_gf, const int64_t* gl, const int64_t* gc, int ng, int d,
                       int32_t* cmc_sum, double* ap, int32_t* valid);

/* ---- multi-GPU exchange: one process per GPU, collectives directly on librccl (RCCL over xGMI) -----
 * The path shards with ONE exchange step: crops are split contiguous-by-index, every rank embeds its shard, one all-gather of
 * the [n_local][512] fp32 embeddings follows, each rank computes its row block of the distance matrix.  A fixed gallery is
 * sharded by rows - the reference's own faiss.IndexShards pattern (reid/faiss_utils.py:121-135: shard, search, merge).
 * Rank 0 creates the id, the host distributes its 128 bytes (any channel: a TCP store, MPI, a file), every rank calls
 * reid_comm_init.  Without a communicator (or world == 1) every collective below degrades to the local copy. */
#define REID_COMM_ID_BYTES 128
int reid_comm_unique_id(void* id128);                                        /* ncclGetUniqueId */
int reid_comm_init(reid_ctx* ctx, int rank, int world, const void* id128);   /* ncclCommInitRank on the context's device */
int reid_comm_info(reid_ctx* ctx, int* rank, int* world);
int reid_comm_destroy(reid_ctx* ctx);
/* d_recv[r * bytes ..] = rank r's d_send (bytes per rank equal on all ranks); enqueued on the context's stream, no sync */
int reid_allgather_dev(reid_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank);
/* ragged row blocks: rank r contributes n_local rows of row_bytes; d_out = all rows in rank order, counts_host[world] (may be
 * NULL) the per-rank row counts, *n_total their sum.  Synchronises the stream once (the counts are needed on the host). */
int reid_allgather_rows_dev(reid_ctx* ctx, const void* d_local, int n_local, size_t row_bytes, void* d_out,
                            int32_t* counts_host, int* n_total);
/* small host-side reduction over the ranks, op 0 = sum, 1 = max, count <= 64; also the job's barrier.  Synchronises. */
int reid_allreduce_f64(reid_ctx* ctx, double* inout_host, int count, int op);
/* squared-L2 k-NN over a gallery whose ROWS are sharded across the ranks (index_init_gpu's IndexShards, faiss_utils.py:121-135):
 * d_xq [nq][d] identical on every rank, d_xb_local [nb_local][d] this rank's rows, index_base their first global row.
 * Same (d_D, d_I) on every rank, equal to a single-process reid_knn_dev over the whole gallery (ties -> lowest global row). */
int reid_knn_gallery_sharded_dev(reid_ctx* ctx, const float* d_xq, int nq, const float* d_xb_local, int nb_local,
                                 int index_base, int d, int k, float* d_D, int32_t* d_I);

/* ---- single operators (unit tests and reuse by other backbones) --------------------------- */
/* NHWC convolution, weights [Cout][R][S][Cin], optional per-channel scale/shift, residual (same shape as out), ReLU */
int reid_conv2d_nhwc(reid_ctx* ctx, const float* x, int n, int h, int w, int cin, const float* wgt, int cout, int r,
                     int s, int stride, int pad, const float* scale, const float* shift, const float* residual,
                     int relu, float* out);
/* C[m][n] = A[m][k] . B[n][k]^T (+ bias[n]) */
int reid_gemm_nt(reid_ctx* ctx, const float* a, int m, const float* b, int n, int k, const float* bias, float* c);

#ifdef __cplusplus
}
#endif
#endif /* REID_HIP_H */
