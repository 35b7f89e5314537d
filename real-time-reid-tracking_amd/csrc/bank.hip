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

// Small updates (a tracking frame: <= 128 samples) carry their metadata as kernel arguments: no upload, no synchronisation.
struct MetaArgs { int32_t v[5 * 128]; };   // src row | slot | pos | count slot | count value
__global__ __launch_bounds__(256) void bank_write_args_kernel(const float* __restrict__ src, const MetaArgs a, int nw, int nc, int budget,
                                                              int d, float* __restrict__ feat, float* __restrict__ sq,
                                                              int32_t* __restrict__ count) {
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    if (i == nw) {   // the block after the samples: new sample counts of the touched tracks (same launch, one fewer in a frame)
        if (tid < nc) count[a.v[384 + tid]] = a.v[512 + tid];
        return;
    }
    const int slot = a.v[128 + i], pos = a.v[256 + i];
    const float* s = src + (long long)a.v[i] * d;
    float* o = feat + ((long long)slot * budget + pos) * d;
    float acc = 0.f;
    for (int k = tid; k < d; k += 256) {
        const float v = s[k];
        o[k] = v;
        acc += v * v;
    }
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) sq[(long long)slot * budget + pos] = (red[0] + red[1]) + (red[2] + red[3]);
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

// d = 512 (every backbone of this path): no LDS in the sample loop.  Lane l keeps elements 8 l .. 8 l + 7 of the block's 16
// detections in 128 registers; a sample costs a wave two coalesced 16-byte loads per lane, 128 FMAs and a 17-shuffle butterfly
// that leaves the 16 dot products one per lane (the kernel above spends 128 LDS reads and 96 shuffles per sample: 50 us for a
// 40 x 30 frame, as long as two layer-4 convolutions).  8 waves share a track's samples.
constexpr int NW5 = 8;
__device__ __forceinline__ float butterfly16(float (&v)[16], int lane) {
    // after step s the lane holds 16 >> s sums over 2^s lanes; detection index j(lane) = 8 b0 + 4 b1 + 2 b2 + b3
    float r8[8], r4[4], r2[2];
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) r8[i] = (b0 ? v[i + 8] : v[i]) + __shfl_xor(b0 ? v[i] : v[i + 8], 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) r4[i] = (b1 ? r8[i + 4] : r8[i]) + __shfl_xor(b1 ? r8[i] : r8[i + 4], 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) r2[i] = (b2 ? r4[i + 2] : r4[i]) + __shfl_xor(b2 ? r4[i] : r4[i + 2], 4);
    float r = (b3 ? r2[1] : r2[0]) + __shfl_xor(b3 ? r2[0] : r2[1], 8);
    r += __shfl_xor(r, 16);
    r += __shfl_xor(r, 32);
    return r;
}
__global__ __launch_bounds__(NW5 * 64) void bank_cost512_kernel(const float* __restrict__ feat, const float* __restrict__ sq,
                                                                const int32_t* __restrict__ count, int budget,
                                                                const int32_t* __restrict__ slots, const float* __restrict__ dets,
                                                                int m, int metric, float gate, float* __restrict__ out) {
    constexpr int D = 512;
    __shared__ float best_sh[NW5][DT];
    const int t = blockIdx.x, j0 = blockIdx.y * DT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nj = m - j0 < DT ? m - j0 : DT;
    const int jl = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);   // this lane's detection
    float dv[DT][8];
    float tmp[16];
#pragma unroll
    for (int j = 0; j < DT; ++j) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        if (j < nj) {
            a = *(const float4*)(dets + (long long)(j0 + j) * D + lane * 8);
            c = *(const float4*)(dets + (long long)(j0 + j) * D + lane * 8 + 4);
        }
        dv[j][0] = a.x; dv[j][1] = a.y; dv[j][2] = a.z; dv[j][3] = a.w;
        dv[j][4] = c.x; dv[j][5] = c.y; dv[j][6] = c.z; dv[j][7] = c.w;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += dv[j][e] * dv[j][e];
        tmp[j] = s;
    }
    const float dsq = butterfly16(tmp, lane);       // |det jl|^2
    const int slot = slots[t];
    const int cnt = count[slot];
    const float* base = feat + (long long)slot * budget * D + lane * 8;
    float best = INFINITY;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    if (wave < cnt) {
        ra = *(const float4*)(base + (long long)wave * D);
        rb = *(const float4*)(base + (long long)wave * D + 4);
    }
    for (int s = wave; s < cnt; s += NW5) {
        const float rv[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
        const float ssq = sq[(long long)slot * budget + s];
        if (s + NW5 < cnt) {                         // next sample's row is in flight during this one's arithmetic
            ra = *(const float4*)(base + (long long)(s + NW5) * D);
            rb = *(const float4*)(base + (long long)(s + NW5) * D + 4);
        }
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a += rv[e] * dv[j][e];
            tmp[j] = a;
        }
        const float dot = butterfly16(tmp, lane);
        float c;
        if (metric == 0) c = 1.f - dot / (sqrtf(ssq) * sqrtf(dsq));
        else c = fmaxf(ssq + dsq - 2.f * dot, 0.f);
        best = fminf(best, c);
    }
    if (lane < 16) best_sh[wave][jl] = best;
    __syncthreads();
    if (tid < nj) {
        float c = best_sh[0][tid];
#pragma unroll
        for (int w = 1; w < NW5; ++w) c = fminf(c, best_sh[w][tid]);
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
// src_rows (optional): sample i is row src_rows[i] of d_feats instead of row i
static int bank_update_impl(reid_ctx* ctx, reid_bank* b, const float* d_feats, const int32_t* slots, int n,
                            const int32_t* src_rows = nullptr) {
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
        const int src = src_rows ? src_rows[i] : i;
        if (it != where.end()) {
            row[it->second] = src;
        } else {
            where[key] = (int)row.size();
            row.push_back(src);
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
    if (nw <= 128 && nc <= 128) {
        MetaArgs a;
        memcpy(a.v, row.data(), nw * 4);
        memcpy(a.v + 128, slot.data(), nw * 4);
        memcpy(a.v + 256, pos.data(), nw * 4);
        memcpy(a.v + 384, cs.data(), nc * 4);
        memcpy(a.v + 512, cv.data(), nc * 4);
        prof_begin(ctx, REID_K_SELECT, 0, 8.0 * nw * b->d);
        hipLaunchKernelGGL(bank_write_args_kernel, dim3(nw + 1), dim3(256), 0, ctx->stream, d_feats, a, nw, nc, b->budget, b->d, b->feat, b->sq,
                           b->count);
        LAUNCH_CHECK();
        prof_end(ctx);
        return REID_OK;
    }
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

// The match stream (opt-in, reid_frame_match_stream): cost and update stages run there, ordered against the forwards by two events per
// slot.  While a stage is queued the context's stream pointer is swapped, so the launchers it calls need not know.  The bank's other
// entry points (host / device features in, costs out) run there too, so that every access to a bank is ordered on ONE stream: `join`
// puts them behind what the compute stream holds so far (the caller's device operands), `rejoin` makes the compute stream wait for
// them (device results the caller goes on to use).
struct MatchStream {
    reid_ctx* ctx;
    hipStream_t saved;
    bool rejoin;
    explicit MatchStream(reid_ctx* c, bool join = false, bool rejoin_ = false) : ctx(c), saved(c->stream), rejoin(rejoin_) {
        if (!on()) return;
        if (join && hipEventRecord(c->join_ev, saved) == hipSuccess) hipStreamWaitEvent(c->match_stream, c->join_ev, 0);
        c->stream = c->match_stream;
    }
    ~MatchStream() {
        ctx->stream = saved;
        if (on() && rejoin && hipEventRecord(ctx->join_ev, ctx->match_stream) == hipSuccess) hipStreamWaitEvent(saved, ctx->join_ev, 0);
    }
    bool on() const { return ctx->match_async && ctx->match_stream; }
};

extern "C" int reid_bank_update_dev(reid_ctx* ctx, reid_bank* b, const float* d_feats, const int32_t* slots, int n) {
    ARG_CHECK(ctx && b && b->ctx == ctx && n >= 0 && (n == 0 || (d_feats && slots)));
    CTX_GUARD(ctx);
    MatchStream ms(ctx, true);
    return bank_update_impl(ctx, b, d_feats, slots, n);
}

extern "C" int reid_bank_update(reid_ctx* ctx, reid_bank* b, const float* feats, const int32_t* slots, int n) {
    ARG_CHECK(ctx && b && b->ctx == ctx && n >= 0 && (n == 0 || (feats && slots)));
    CTX_GUARD(ctx);
    if (n == 0) return REID_OK;
    MatchStream ms(ctx, true);
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
    MatchStream ms(ctx, true);
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

static int bank_cost_launch(reid_ctx* ctx, reid_bank* b, const int32_t* d_slots, int t, const float* d_dets, int m, int metric,
                            float max_dist, float* d_out) {
    if (b->d == 512 && ctx->bank_fast) {
        prof_begin(ctx, REID_K_SELECT, 2.0 * t * m * b->budget * b->d, 4.0 * ((double)t * b->budget * b->d + (double)m * b->d));
        hipLaunchKernelGGL(bank_cost512_kernel, dim3(t, (m + DT - 1) / DT), dim3(NW5 * 64), 0, ctx->stream, b->feat, b->sq, b->count,
                           b->budget, d_slots, d_dets, m, metric, max_dist, d_out);
        LAUNCH_CHECK();
        prof_end(ctx);
        return REID_OK;
    }
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

static int bank_cost_impl(reid_ctx* ctx, reid_bank* b, const int32_t* slots, int t, const float* d_dets, int m, int metric,
                          float max_dist, float* d_out) {
    int32_t* d_slots;
    REID_TRY(ctx_ws(ctx, "bank.slots", (size_t)t * 4, (void**)&d_slots));
    for (int i = 0; i < t; ++i) ARG_CHECK(slots[i] >= 0 && slots[i] < b->max_tracks);
    HIP_TRY(hipMemcpyAsync(d_slots, slots, (size_t)t * 4, hipMemcpyHostToDevice, ctx->stream));
    return bank_cost_launch(ctx, b, d_slots, t, d_dets, m, metric, max_dist, d_out);
}

// cost[t][m]; metric REID_METRIC_COS (1 - cosine) or REID_METRIC_L2SQR; max_dist < 0: raw, else cost > max_dist -> max_dist + 1e-5
extern "C" int reid_bank_cost_dev(reid_ctx* ctx, reid_bank* b, const int32_t* slots, int t, const float* d_dets, int m,
                                  int metric, float max_dist, float* d_out) {
    ARG_CHECK(ctx && b && b->ctx == ctx && t >= 0 && m >= 0);
    CTX_GUARD(ctx);
    ARG_CHECK(metric == REID_METRIC_COS || metric == REID_METRIC_L2SQR);
    if (t == 0 || m == 0) return REID_OK;
    ARG_CHECK(slots && d_dets && d_out);
    MatchStream ms(ctx, true, true);
    return bank_cost_impl(ctx, b, slots, t, d_dets, m, metric == REID_METRIC_COS ? 0 : 1, max_dist, d_out);
}

extern "C" int reid_bank_cost(reid_ctx* ctx, reid_bank* b, const int32_t* slots, int t, const float* dets, int m, int metric,
                              float max_dist, float* out) {
    ARG_CHECK(ctx && b && b->ctx == ctx && t >= 0 && m >= 0);
    CTX_GUARD(ctx);
    ARG_CHECK(metric == REID_METRIC_COS || metric == REID_METRIC_L2SQR);
    if (t == 0 || m == 0) return REID_OK;
    ARG_CHECK(slots && dets && out);
    MatchStream ms(ctx, true);
    float *d_dets, *d_out;
    REID_TRY(ctx_ws(ctx, "bank.dets", (size_t)m * b->d * 4, (void**)&d_dets));
    REID_TRY(ctx_ws(ctx, "bank.out", (size_t)t * m * 4, (void**)&d_out));
    HIP_TRY(hipMemcpyAsync(d_dets, dets, (size_t)m * b->d * 4, hipMemcpyHostToDevice, ctx->stream));
    REID_TRY(bank_cost_impl(ctx, b, slots, t, d_dets, m, metric == REID_METRIC_COS ? 0 : 1, max_dist, d_out));
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)t * m * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Frame pipeline: what DeepSort.update ([external] deep_sort.py: _get_features -> Tracker.update -> metric.distance /
// iou_cost -> metric.partial_fit) asks of this path each frame, as three stages with ONE stream synchronisation:
//   submit : crops -> device, resize + normalise, forward                     (asynchronous)
//   cost   : appearance cost (gated) + DIoU cost, results -> pinned staging    (asynchronous, ends in an event)
//   fetch  : wait for that event only, hand the results over                   (the frame's one wait)
//   update : partial_fit from the embeddings still on the device               (asynchronous, metadata as kernel arguments)
// Call order of a stream: cost(f), submit(f+1), fetch(f), <assignment on the host>, update(f): the stream then holds
// forward(f) | cost(f) | forward(f+1) | update(f) | cost(f+1) ..., the host waits for cost(f) only and the device runs
// forward(f+1) under the host's assignment and packing work.
// Two frame slots own their device-side crops and embeddings, so frame f+1 is uploaded and embedded while the host still
// runs the assignment of frame f.  Small inputs go through pinned staging buffers: a hipMemcpyAsync from pageable memory
// would block the caller until everything queued before it has finished.
extern "C" int reid_frame_match_stream(reid_ctx* ctx, int on) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->match_stream) HIP_TRY(hipStreamSynchronize(ctx->match_stream));
    if (on && !ctx->match_stream) {
        // HIP multiplexes a process's streams onto a few hardware queues per priority level (4 by default): with several contexts alive
        // the match stream could land in the queue of this context's compute stream and every cost stage would sit behind the forward
        // it is meant to run beside (seen in bench.py: 1.44 k frames/s instead of 1.95 k).  High priority puts it in a queue pool of its
        // own - and lets its small kernels in between the forward's.
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&ctx->match_stream, hipStreamNonBlocking, greatest));
        HIP_TRY(hipEventCreateWithFlags(&ctx->join_ev, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&ctx->fwd_ev[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ctx->match_ev[i], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ctx->match_ev[i], ctx->match_stream));
        }
    }
    ctx->match_async = on ? 1 : 0;
    return REID_OK;
}

extern "C" int reid_frame_submit(reid_ctx* ctx, int slot, const uint8_t* packed, const int64_t* offsets, const int32_t* hw, int m) {
    ARG_CHECK(ctx && (slot == 0 || slot == 1) && m >= 0 && (m == 0 || (packed && offsets && hw)));
    CTX_ENTER(ctx);
    if (ctx->frame_pending[slot]) {   // resubmitted without reid_frame_fetch: its staging is still in use
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->match_stream) HIP_TRY(hipStreamSynchronize(ctx->match_stream));
    }
    if (ctx->match_async && ctx->match_stream)   // the slot's buffers are rewritten below: behind the last cost / update stage that read them
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->match_ev[slot], 0));
    ctx->frame_has[slot] = 0;
    ctx->frame_pending[slot] = 0;
    ctx->frame_m[slot] = 0;
    ctx->frame_emb[slot] = nullptr;
    if (m == 0) return REID_OK;
    char* pin;
    const std::string tag = slot ? "frame1" : "frame0";
    REID_TRY(ctx_pinned(ctx, (tag + ".meta").c_str(), (size_t)m * 16, (void**)&pin));
    memcpy(pin, offsets, (size_t)m * 8);
    memcpy(pin + (size_t)m * 8, hw, (size_t)m * 8);
    float* d_emb;
    REID_TRY(embed_ragged_enqueue(ctx, tag.c_str(), packed, (const int64_t*)pin, (const int32_t*)(pin + (size_t)m * 8), m, &d_emb, nullptr,
                                  ctx->side_copy != 0 && ctx->stream != nullptr));
    ctx->frame_m[slot] = m;
    ctx->frame_emb[slot] = d_emb;
    ctx->frame_pending[slot] = 1;
    if (ctx->match_async && ctx->match_stream) HIP_TRY(hipEventRecord(ctx->fwd_ev[slot], ctx->stream));
    return REID_OK;
}

// Cost stage for `groups` camera streams batched into one frame slot (reid_frame_cost is the one-group case): the slot's m
// detections are the cameras' detections one camera after the other (m_counts[g] each); group g's tracks are matched against ITS
// detections only, on ITS bank.  Host arrays are the groups' concatenations; outputs are the groups' t_g x m_g blocks, concatenated.
extern "C" int reid_frame_cost_groups(reid_ctx* ctx, int slot, int groups, reid_bank* const* banks, const int32_t* t_counts,
                                      const int32_t* m_counts, const int32_t* slots, int metric, float max_dist, const double* tracks_t4,
                                      const double* dets_m4, int want_emb) {
    ARG_CHECK(ctx && (slot == 0 || slot == 1) && groups >= 1 && t_counts && m_counts);
    CTX_ENTER(ctx);
    const int m = ctx->frame_m[slot];
    int t = 0, msum = 0;
    size_t tm = 0;
    for (int g = 0; g < groups; ++g) {
        ARG_CHECK(t_counts[g] >= 0 && m_counts[g] >= 0);
        t += t_counts[g];
        msum += m_counts[g];
        tm += (size_t)t_counts[g] * m_counts[g];
    }
    ARG_CHECK(msum == m);
    const bool want_cost = banks && slots && tm > 0, want_iou = tracks_t4 && dets_m4 && tm > 0;
    MatchStream ms(ctx);
    if (ms.on() && m > 0) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->fwd_ev[slot], 0));     // the slot's embeddings
    if (want_cost) {
        ARG_CHECK(metric == REID_METRIC_COS || metric == REID_METRIC_L2SQR);
        for (int g = 0, i = 0; g < groups; ++g) {
            if (t_counts[g] == 0 || m_counts[g] == 0) { i += t_counts[g]; continue; }
            ARG_CHECK(banks[g] && banks[g]->ctx == ctx && banks[g]->d == 512);
            for (int e = i + t_counts[g]; i < e; ++i) ARG_CHECK(slots[i] >= 0 && slots[i] < banks[g]->max_tracks);
        }
    }
    const std::string tag = slot ? "frame1" : "frame0";
    // inputs: [tracks 32 t][dets 32 m][slots 4 t]; outputs: [iou 8 tm][cost 4 tm][emb 2048 m]
    const size_t in_bytes = (size_t)t * 32 + (size_t)m * 32 + (size_t)t * 4;
    const size_t out_bytes = tm * 12 + (size_t)m * 2048;
    char *pin_in, *pin_out, *d_in, *d_out;
    REID_TRY(ctx_pinned(ctx, (tag + ".cin").c_str(), in_bytes + 8, (void**)&pin_in));
    REID_TRY(ctx_pinned(ctx, (tag + ".cout").c_str(), out_bytes + 8, (void**)&pin_out));
    REID_TRY(ctx_ws(ctx, (tag + ".cin").c_str(), in_bytes + 8, (void**)&d_in));
    REID_TRY(ctx_ws(ctx, (tag + ".cout").c_str(), tm * 12 + 8, (void**)&d_out));
    if (want_iou) {
        memcpy(pin_in, tracks_t4, (size_t)t * 32);
        memcpy(pin_in + (size_t)t * 32, dets_m4, (size_t)m * 32);
    }
    if (want_cost) memcpy(pin_in + (size_t)t * 32 + (size_t)m * 32, slots, (size_t)t * 4);
    if (want_iou || want_cost) HIP_TRY(hipMemcpyAsync(d_in, pin_in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    double* d_iou = (double*)d_out;
    float* d_cost = (float*)(d_out + tm * 8);
    const double* d_tracks = (const double*)d_in;
    const double* d_dets = (const double*)(d_in + (size_t)t * 32);
    const int32_t* d_slots = (const int32_t*)(d_in + (size_t)t * 32 + (size_t)m * 32);
    size_t t_off = 0, m_off = 0, tm_off = 0;
    for (int g = 0; g < groups; ++g) {
        const int tg = t_counts[g], mg = m_counts[g];
        if (tg > 0 && mg > 0) {
            if (want_cost)
                REID_TRY(bank_cost_launch(ctx, banks[g], d_slots + t_off, tg, ctx->frame_emb[slot] + m_off * 512, mg, metric == REID_METRIC_COS ? 0 : 1,
                                          max_dist, d_cost + tm_off));
            if (want_iou) REID_TRY(launch_diou_cost(ctx, d_tracks + t_off * 4, tg, d_dets + m_off * 4, mg, d_iou + tm_off, 1));
        }
        t_off += tg;
        m_off += mg;
        tm_off += (size_t)tg * mg;
    }
    if (want_iou && want_cost) HIP_TRY(hipMemcpyAsync(pin_out, d_iou, tm * 12, hipMemcpyDeviceToHost, ctx->stream));   // adjacent: one copy
    else if (want_iou) HIP_TRY(hipMemcpyAsync(pin_out, d_iou, tm * 8, hipMemcpyDeviceToHost, ctx->stream));
    else if (want_cost) HIP_TRY(hipMemcpyAsync(pin_out + tm * 8, d_cost, tm * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (want_emb && m > 0)
        HIP_TRY(hipMemcpyAsync(pin_out + tm * 12, ctx->frame_emb[slot], (size_t)m * 2048, hipMemcpyDeviceToHost, ctx->stream));
    if (!ctx->frame_ev[slot]) HIP_TRY(hipEventCreateWithFlags(&ctx->frame_ev[slot], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ctx->frame_ev[slot], ctx->stream));
    if (ms.on()) HIP_TRY(hipEventRecord(ctx->match_ev[slot], ctx->stream));
    ctx->frame_tm[slot] = tm;
    ctx->frame_has[slot] = (want_iou ? 1 : 0) | (want_cost ? 2 : 0) | (want_emb && m > 0 ? 4 : 0) | 8;
    ctx->frame_out[slot] = pin_out;
    return REID_OK;
}

extern "C" int reid_frame_cost(reid_ctx* ctx, int slot, reid_bank* b, const int32_t* slots, int t, int metric, float max_dist,
                               const double* tracks_t4, const double* dets_m4, int want_emb) {
    ARG_CHECK(ctx && (slot == 0 || slot == 1) && t >= 0);
    const int32_t tc = t, mc = ctx->frame_m[slot];
    reid_bank* const banks[1] = {b};
    return reid_frame_cost_groups(ctx, slot, 1, (b && slots) ? banks : nullptr, &tc, &mc, slots, metric, max_dist, tracks_t4, dets_m4, want_emb);
}

extern "C" int reid_frame_fetch(reid_ctx* ctx, int slot, float* emb, float* cost_tm, double* iou_tm) {
    ARG_CHECK(ctx && (slot == 0 || slot == 1));
    CTX_ENTER(ctx);
    if (!(ctx->frame_has[slot] & 8)) {
        reid_set_error("reid_frame_fetch: no reid_frame_cost pending on slot %d", slot);
        return REID_ERR_STATE;
    }
    HIP_TRY(hipEventSynchronize(ctx->frame_ev[slot]));   // later work of the stream (the next frame's forward) keeps running
    const int has = ctx->frame_has[slot];
    ctx->frame_has[slot] = 0;
    ctx->frame_pending[slot] = 0;
    const size_t tm = ctx->frame_tm[slot];
    const char* pin_out = ctx->frame_out[slot];
    ARG_CHECK((!iou_tm || (has & 1)) && (!cost_tm || (has & 2)) && (!emb || (has & 4) || ctx->frame_m[slot] == 0));
    if (iou_tm) memcpy(iou_tm, pin_out, tm * 8);
    if (cost_tm) memcpy(cost_tm, pin_out + tm * 8, tm * 4);
    if (emb && (has & 4)) memcpy(emb, pin_out + tm * 12, (size_t)ctx->frame_m[slot] * 2048);
    return ctx_fault_status(ctx);   // the frame's forward has completed: a fault it raised is reported with its results
}

extern "C" int reid_frame_update(reid_ctx* ctx, int slot, reid_bank* b, const int32_t* rows, const int32_t* slots, int n) {
    ARG_CHECK(ctx && (slot == 0 || slot == 1) && b && b->ctx == ctx && b->d == 512 && n >= 0 && (n == 0 || (rows && slots)));
    CTX_GUARD(ctx);
    if (n == 0) return REID_OK;
    for (int i = 0; i < n; ++i) ARG_CHECK(rows[i] >= 0 && rows[i] < ctx->frame_m[slot]);
    MatchStream ms(ctx);      // (behind the slot's cost stage on the same stream, which waited for its forward)
    REID_TRY(bank_update_impl(ctx, b, ctx->frame_emb[slot], slots, n, rows));
    if (ms.on()) HIP_TRY(hipEventRecord(ctx->match_ev[slot], ctx->stream));
    return REID_OK;
}
