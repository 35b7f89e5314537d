// DeepSORT appearance metric with the per-track feature bank kept in HBM.
//
// What it replaces ([external] deep_sort/sort/nn_matching.py, the immediate consumer of Extractor.__call__ every frame;
// parameters from the reference's modification_deepsort/deep_sort.yaml:3,9  MAX_DIST 0.15, NN_BUDGET 100):
//   NearestNeighborDistanceMetric.partial_fit : samples[target].append(feature); keep the last `budget`
//   NearestNeighborDistanceMetric.distance    : cost[i, :] = min over samples[target_i] of metric(sample, detections)
//   _nn_cosine_distance    = (1 - a_hat . b_hat).min(axis=0)
//   _nn_euclidean_distance = max(0, clip(|a|^2 + |b|^2 - 2 a.b, 0, inf).min(axis=0))
//   linear_assignment.min_cost_matching : cost[cost > max_distance] = max_distance + 1e-5   (optional, fused)
// The reference runs a Python loop over tracks with one small numpy GEMM each; here the bank is a ring buffer
// [slot][budget][d] fp32 that never leaves the device, an update is one row-copy kernel, and the T x M cost matrix is one
// launch: block = (track, 16 detections), the 16 detection rows sit in LDS, each wave streams bank rows once (coalesced)
// against all 16 and keeps the running minimum in registers.  The minimum is order-free, so ring order does not matter.
#include "reid_internal.h"
#include <string.h>
#include <utility>

struct reid_bank {
    reid_ctx* ctx;
    int max_tracks, budget, d;
    float* feat;      // [max_tracks][budget][d]
    float* sq;        // [max_tracks][budget]  squared norms
    int32_t* count;   // [max_tracks] device copy of min(total, budget)
    std::vector<int32_t> h_total;   // host mirror: samples ever written per slot
};

namespace {

constexpr int DT = 16;   // detections per block

__global__ __launch_bounds__(256) void bank_write_kernel(const float* __restrict__ src, const int32_t* __restrict__ src_row,
                                                         const int32_t* __restrict__ dst_slot,
                                                         const int32_t* __restrict__ dst_pos, int budget, int d,
                                                         float* __restrict__ feat, float* __restrict__ sq) {
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float* s = src + (long long)src_row[i] * d;
    float* o = feat + ((long long)dst_slot[i] * budget + dst_pos[i]) * d;
    float acc = 0.f;
    for (int k = tid; k < d; k += 256) {
        const float v = s[k];
        o[k] = v;
        acc += v * v;
    }
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) sq[(long long)dst_slot[i] * budget + dst_pos[i]] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void bank_set_count_kernel(const int32_t* __restrict__ slots, const int32_t* __restrict__ values, int n,
                                      int32_t* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) count[slots[i]] = values[i];
}

// metric 0: cosine, 1: squared euclidean.  gate < 0: no clamp.  NW waves per block share a track's samples: a tracking frame
// launches only tracks x ceil(dets / 16) blocks (80 for 40 x 30), so the per-block latency - budget / NW samples per wave - is
// the kernel's time; 16 waves instead of 4 cut it ~4x.
constexpr int NW = 16;
__global__ __launch_bounds__(NW * 64) void bank_cost_kernel(const float* __restrict__ feat, const float* __restrict__ sq,
                                                        const int32_t* __restrict__ count, int budget, int d,
                                                        const int32_t* __restrict__ slots, const float* __restrict__ dets,
                                                        int m, int metric, float gate, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float det_lds[];   // [DT][d]
    __shared__ float det_sq[DT];
    __shared__ float best_sh[NW][DT];
    const int t = blockIdx.x, j0 = blockIdx.y * DT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nj = m - j0 < DT ? m - j0 : DT;
    for (int idx = tid; idx < DT * d; idx += NW * 64) {
        const int j = idx / d;
        det_lds[idx] = j < nj ? dets[(long long)j0 * d + idx] : 0.f;
    }
    __syncthreads();
    for (int j = wave; j < DT; j += NW) {
        float a = 0.f;
        for (int k = lane; k < d; k += 64) a += det_lds[j * d + k] * det_lds[j * d + k];
        for (int off = 32; off; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0) det_sq[j] = a;
    }
    __syncthreads();
    const int slot = slots[t];
    const int cnt = count[slot];
    float best[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) best[j] = INFINITY;
    for (int s = wave; s < cnt; s += NW) {
        const float* row = feat + ((long long)slot * budget + s) * d;
        float dot[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) dot[j] = 0.f;
        for (int k = lane; k < d; k += 64) {
            const float v = row[k];
#pragma unroll
            for (int j = 0; j < DT; ++j) dot[j] += v * det_lds[j * d + k];
        }
        const float ssq = sq[(long long)slot * budget + s];
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            float a = dot[j];
            for (int off = 32; off; off >>= 1) a += __shfl_xor(a, off);
            float c;
            if (metric == 0) c = 1.f - a / (sqrtf(ssq) * sqrtf(det_sq[j]));
            else c = fmaxf(ssq + det_sq[j] - 2.f * a, 0.f);
            best[j] = fminf(best[j], c);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < DT; ++j) best_sh[wave][j] = best[j];
    }
    __syncthreads();
    if (tid < nj) {
        float c = best_sh[0][tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) c = fminf(c, best_sh[w][tid]);
        if (cnt == 0) c = gate >= 0.f ? gate + 1e-5f : INFINITY;   // a track without samples matches nothing
        else if (gate >= 0.f && c > gate) c = gate + 1e-5f;
        out[(long long)t * m + j0 + tid] = c;
    }
}

}  // namespace

extern "C" int reid_bank_create(reid_ctx* ctx, int max_tracks, int budget, int d, reid_bank** out) {
    ARG_CHECK(ctx && out && max_tracks >= 1 && budget >= 1 && d >= 1 && d <= 2048);
    CTX_GUARD(ctx);
    ARG_CHECK((double)max_tracks * budget * d * 4.0 < 64e9);
    reid_bank* b = new reid_bank();
    b->ctx = ctx;
    b->max_tracks = max_tracks;
    b->budget = budget;
    b->d = d;
    b->h_total.assign(max_tracks, 0);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc((void**)&b->feat, (size_t)max_tracks * budget * d * 4));
    HIP_TRY(hipMalloc((void**)&b->sq, (size_t)max_tracks * budget * 4));
    HIP_TRY(hipMalloc((void**)&b->count, (size_t)max_tracks * 4));
    HIP_TRY(hipMemsetAsync(b->count, 0, (size_t)max_tracks * 4, ctx->stream));
    *out = b;
    return REID_OK;
}

extern "C" int reid_bank_destroy(reid_bank* b) {
    if (!b) return REID_OK;
    DeviceGuard _dev_guard(b->ctx->device);
    hipFree(b->feat);
    hipFree(b->sq);
    hipFree(b->count);
    delete b;
    return REID_OK;
}

extern "C" int reid_bank_count(reid_bank* b, int slot, int* out) {
    ARG_CHECK(b && out && slot >= 0 && slot < b->max_tracks);
    *out = b->h_total[slot] < b->budget ? b->h_total[slot] : b->budget;
    return REID_OK;
}

// partial_fit: sample i (row i of feats) is appended to track slots[i], in order; only the last `budget` per track survive.
static int bank_update_impl(reid_ctx* ctx, reid_bank* b, const float* d_feats, const int32_t* slots, int n) {
    if (n == 0) return REID_OK;
    std::vector<int32_t> row, slot, pos, cs, cv;
    // positions in call order; a later sample landing on the same (slot, pos) replaces the earlier one
    std::map<std::pair<int, int>, int> where;
    std::vector<int32_t> total = b->h_total;
    for (int i = 0; i < n; ++i) {
        const int s = slots[i];
        ARG_CHECK(s >= 0 && s < b->max_tracks);
        const int p = total[s] % b->budget;
        total[s]++;
        auto key = std::make_pair(s, p);
        auto it = where.find(key);
        if (it != where.end()) {
            row[it->second] = i;
        } else {
            where[key] = (int)row.size();
            row.push_back(i);
            slot.push_back(s);
            pos.push_back(p);
        }
    }
    for (int i = 0; i < n; ++i) {
        const int s = slots[i];
        if (total[s] != b->h_total[s]) {
            // wrap guard for very long-lived tracks: keep total in [budget, 2*budget) once the ring is full
            if (total[s] >= 2 * b->budget) total[s] = b->budget + total[s] % b->budget;
            cs.push_back(s);
            cv.push_back(total[s] < b->budget ? total[s] : b->budget);
            b->h_total[s] = total[s];
        }
    }
    const int nw = (int)row.size(), nc = (int)cs.size();
    int32_t* meta;
    REID_TRY(ctx_ws(ctx, "bank.meta", (size_t)(3 * nw + 2 * nc) * 4, (void**)&meta));
    std::vector<int32_t> h(3 * nw + 2 * nc);
    memcpy(h.data(), row.data(), nw * 4);
    memcpy(h.data() + nw, slot.data(), nw * 4);
    memcpy(h.data() + 2 * nw, pos.data(), nw * 4);
    memcpy(h.data() + 3 * nw, cs.data(), nc * 4);
    memcpy(h.data() + 3 * nw + nc, cv.data(), nc * 4);
    HIP_TRY(hipMemcpyAsync(meta, h.data(), h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // h is a stack-lifetime buffer
    prof_begin(ctx, REID_K_SELECT, 0, 8.0 * nw * b->d);
    hipLaunchKernelGGL(bank_write_kernel, dim3(nw), dim3(256), 0, ctx->stream, d_feats, meta, meta + nw, meta + 2 * nw,
                       b->budget, b->d, b->feat, b->sq);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(bank_set_count_kernel, dim3((nc + 255) / 256), dim3(256), 0, ctx->stream, meta + 3 * nw,
                       meta + 3 * nw + nc, nc, b->count);
    LAUNCH_CHECK();
    prof_end(ctx);
    return REID_OK;
}

extern "C" int reid_bank_update_dev(reid_ctx* ctx, reid_bank* b, const float* d_feats, const int32_t* slots, int n) {
    ARG_CHECK(ctx && b && b->ctx == ctx && n >= 0 && (n == 0 || (d_feats && slots)));
    CTX_GUARD(ctx);
    return bank_update_impl(ctx, b, d_feats, slots, n);
}

extern "C" int reid_bank_update(reid_ctx* ctx, reid_bank* b, const float* feats, const int32_t* slots, int n) {
    ARG_CHECK(ctx && b && b->ctx == ctx && n >= 0 && (n == 0 || (feats && slots)));
    CTX_GUARD(ctx);
    if (n == 0) return REID_OK;
    float* d_f;
    REID_TRY(ctx_ws(ctx, "bank.in", (size_t)n * b->d * 4, (void**)&d_f));
    HIP_TRY(hipMemcpyAsync(d_f, feats, (size_t)n * b->d * 4, hipMemcpyHostToDevice, ctx->stream));
    return bank_update_impl(ctx, b, d_f, slots, n);
}

// forget tracks (the reference drops every target not in `active_targets`): their slots can be handed out again
extern "C" int reid_bank_clear(reid_ctx* ctx, reid_bank* b, const int32_t* slots, int n) {
    ARG_CHECK(ctx && b && b->ctx == ctx && n >= 0 && (n == 0 || slots));
    CTX_GUARD(ctx);
    if (n == 0) return REID_OK;
    std::vector<int32_t> h(2 * n, 0);
    for (int i = 0; i < n; ++i) {
        ARG_CHECK(slots[i] >= 0 && slots[i] < b->max_tracks);
        h[i] = slots[i];
        b->h_total[slots[i]] = 0;
    }
    int32_t* meta;
    REID_TRY(ctx_ws(ctx, "bank.meta", (size_t)2 * n * 4, (void**)&meta));
    HIP_TRY(hipMemcpyAsync(meta, h.data(), h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    hipLaunchKernelGGL(bank_set_count_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, meta, meta + n, n, b->count);
    LAUNCH_CHECK();
    return REID_OK;
}

static int bank_cost_impl(reid_ctx* ctx, reid_bank* b, const int32_t* slots, int t, const float* d_dets, int m, int metric,
                          float max_dist, float* d_out) {
    int32_t* d_slots;
    REID_TRY(ctx_ws(ctx, "bank.slots", (size_t)t * 4, (void**)&d_slots));
    for (int i = 0; i < t; ++i) ARG_CHECK(slots[i] >= 0 && slots[i] < b->max_tracks);
    HIP_TRY(hipMemcpyAsync(d_slots, slots, (size_t)t * 4, hipMemcpyHostToDevice, ctx->stream));
    const size_t sh = (size_t)DT * b->d * 4;
    if (sh > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)bank_cost_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    prof_begin(ctx, REID_K_SELECT, 2.0 * t * m * b->budget * b->d, 4.0 * ((double)t * b->budget * b->d + (double)m * b->d));
    hipLaunchKernelGGL(bank_cost_kernel, dim3(t, (m + DT - 1) / DT), dim3(NW * 64), sh, ctx->stream, b->feat, b->sq, b->count,
                       b->budget, b->d, d_slots, d_dets, m, metric, max_dist, d_out);
    LAUNCH_CHECK();
    prof_end(ctx);
    return REID_OK;
}

// cost[t][m]; metric REID_METRIC_COS (1 - cosine) or REID_METRIC_L2SQR; max_dist < 0: raw, else cost > max_dist -> max_dist + 1e-5
extern "C" int reid_bank_cost_dev(reid_ctx* ctx, reid_bank* b, const int32_t* slots, int t, const float* d_dets, int m,
                                  int metric, float max_dist, float* d_out) {
    ARG_CHECK(ctx && b && b->ctx == ctx && t >= 0 && m >= 0);
    CTX_GUARD(ctx);
    ARG_CHECK(metric == REID_METRIC_COS || metric == REID_METRIC_L2SQR);
    if (t == 0 || m == 0) return REID_OK;
    ARG_CHECK(slots && d_dets && d_out);
    return bank_cost_impl(ctx, b, slots, t, d_dets, m, metric == REID_METRIC_COS ? 0 : 1, max_dist, d_out);
}

extern "C" int reid_bank_cost(reid_ctx* ctx, reid_bank* b, const int32_t* slots, int t, const float* dets, int m, int metric,
                              float max_dist, float* out) {
    ARG_CHECK(ctx && b && b->ctx == ctx && t >= 0 && m >= 0);
    CTX_GUARD(ctx);
    ARG_CHECK(metric == REID_METRIC_COS || metric == REID_METRIC_L2SQR);
    if (t == 0 || m == 0) return REID_OK;
    ARG_CHECK(slots && dets && out);
    float *d_dets, *d_out;
    REID_TRY(ctx_ws(ctx, "bank.dets", (size_t)m * b->d * 4, (void**)&d_dets));
    REID_TRY(ctx_ws(ctx, "bank.out", (size_t)t * m * 4, (void**)&d_out));
    HIP_TRY(hipMemcpyAsync(d_dets, dets, (size_t)m * b->d * 4, hipMemcpyHostToDevice, ctx->stream));
    REID_TRY(bank_cost_impl(ctx, b, slots, t, d_dets, m, metric == REID_METRIC_COS ? 0 : 1, max_dist, d_out));
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)t * m * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}
