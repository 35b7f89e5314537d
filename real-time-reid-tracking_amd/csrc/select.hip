// Selection kernels of the matching side: row arg-min, brute-force top-k, rank counting for CMC/mAP, DIoU.
#include "reid_internal.h"
#include <math.h>

namespace {

// order-preserving float -> uint key (total order, -0 < +0), packed with the index so that a 64-bit min gives
// "smallest value, then lowest index"
__device__ __forceinline__ unsigned long long pack_key(float v, int idx) {
    unsigned int u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned int)idx;
}
__device__ __forceinline__ float unpack_val(unsigned long long k) {
    unsigned int u = (unsigned int)(k >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(v, o);
        v = other < v ? other : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long block_min_u64(unsigned long long v, unsigned long long* sh) {
    v = wave_min_u64(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    unsigned long long r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = sh[w] < r ? sh[w] : r;
    return r;
}

// one block per row; NaNs sort last (their key is the largest)
__global__ __launch_bounds__(256) void argmin_rows_kernel(const float* __restrict__ dist, int n, long long ld,
                                                          int32_t* __restrict__ idx, float* __restrict__ val) {
    __shared__ unsigned long long sh[4];
    const float* row = dist + (long long)blockIdx.x * ld;
    unsigned long long best = ~0ull;
    for (int j = threadIdx.x; j < n; j += 256) {
        const unsigned long long k = pack_key(row[j], j);
        best = k < best ? k : best;
    }
    best = block_min_u64(best, sh);
    if (threadIdx.x == 0) {
        idx[blockIdx.x] = (int32_t)(best & 0xffffffffu);
        if (val) val[blockIdx.x] = unpack_val(best);
    }
}

// k smallest of each row, ascending, ties -> lowest index.
// One pass over the row: every thread keeps the FOUR smallest keys of the elements it visits, in registers; then k rounds of
// "block minimum of the threads' heads", the winner pops its list.  A row's global top-k lies in its threads' top-4 lists unless
// one thread holds five or more of them (elements are dealt round-robin: probability ~1e-6 per row at k = 20, n = 16 k); a
// thread whose list runs dry although it saw more than four elements raises a flag and the row is redone by the exact
// k-pass scan below.  (The k-pass scan alone read each row k times and spent 1.6 ms of a 2.2 ms Market-size search.)
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ dist, int n, long long ld, int k,
                                                        float* __restrict__ D, int32_t* __restrict__ I) {
    __shared__ unsigned long long sh[4];
    __shared__ int overflow;
    const float* row = dist + (long long)blockIdx.x * ld;
    if (threadIdx.x == 0) overflow = 0;
    unsigned long long t0 = ~0ull, t1 = ~0ull, t2 = ~0ull, t3 = ~0ull;   // ascending
    int seen = 0;
    for (int j = threadIdx.x; j < n; j += 256, ++seen) {
        unsigned long long key = pack_key(row[j], j);
        if (key < t3) {   // insert, keeping the four smallest in order
            unsigned long long a;
            a = key < t0 ? t0 : key; t0 = key < t0 ? key : t0; key = a;
            a = key < t1 ? t1 : key; t1 = key < t1 ? key : t1; key = a;
            a = key < t2 ? t2 : key; t2 = key < t2 ? key : t2; key = a;
            t3 = key < t3 ? key : t3;
        }
    }
    int popped = 0;
    for (int r = 0; r < k; ++r) {
        const unsigned long long best = block_min_u64(t0, sh);
        if (threadIdx.x == 0) {
            if (best == ~0ull) {  // fewer than k candidates
                D[(long long)blockIdx.x * k + r] = INFINITY;
                I[(long long)blockIdx.x * k + r] = -1;
            } else {
                D[(long long)blockIdx.x * k + r] = unpack_val(best);
                I[(long long)blockIdx.x * k + r] = (int32_t)(best & 0xffffffffu);
            }
        }
        if (t0 == best && best != ~0ull) {   // keys are unique (they carry the index): exactly one thread pops
            t0 = t1; t1 = t2; t2 = t3; t3 = ~0ull;
            if (++popped == 4 && seen > 4) overflow = 1;   // this thread may hold further members of the top-k
        }
    }
    __syncthreads();
    if (!overflow) return;
    // exact fallback: k rounds of "smallest key greater than the last one" over the whole row
    unsigned long long last = 0;
    bool first = true;
    for (int r = 0; r < k; ++r) {
        unsigned long long best = ~0ull;
        for (int j = threadIdx.x; j < n; j += 256) {
            const unsigned long long key = pack_key(row[j], j);
            if ((first || key > last) && key < best) best = key;
        }
        best = block_min_u64(best, sh);
        if (threadIdx.x == 0) {
            if (best == ~0ull) {
                D[(long long)blockIdx.x * k + r] = INFINITY;
                I[(long long)blockIdx.x * k + r] = -1;
            } else {
                D[(long long)blockIdx.x * k + r] = unpack_val(best);
                I[(long long)blockIdx.x * k + r] = (int32_t)(best & 0xffffffffu);
            }
        }
        last = best;
        first = false;
    }
}

// reid/evaluate.py:55-105 without sorting.  For query q: good = same pid & other cam, junk = pid -1 or same pid & cam.
// Order = descending score (ties: higher gallery index first = reversed stable ascending argsort).  For every good
// item: rank among non-junk items and rank among good items, then AP in the reference's own summation order (fp64).
constexpr int MAX_GOOD = 2048;
__global__ __launch_bounds__(256) void rank_eval_kernel(const float* __restrict__ score, int ng, long long ld,
                                                        const long long* __restrict__ ql, const long long* __restrict__ qc,
                                                        const long long* __restrict__ gl, const long long* __restrict__ gc,
                                                        int32_t* __restrict__ first_hist, double* __restrict__ ap,
                                                        int32_t* __restrict__ valid) {
    __shared__ int good[MAX_GOOD];
    __shared__ int rank_all[MAX_GOOD];
    __shared__ int rank_good[MAX_GOOD];
    __shared__ int slot_rank[MAX_GOOD];
    __shared__ int ngood_sh;
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* row = score + (long long)q * ld;
    const long long pid = ql[q], cam = qc[q];
    if (tid == 0) ngood_sh = 0;
    __syncthreads();
    for (int j = tid; j < ng; j += 256) {
        if (gl[j] == pid && gc[j] != cam) {
            const int s = atomicAdd(&ngood_sh, 1);
            if (s < MAX_GOOD) good[s] = j;
        }
    }
    __syncthreads();
    const int ngood = ngood_sh;
    if (ngood == 0 || ngood > MAX_GOOD) {
        if (tid == 0) { ap[q] = 0.0; valid[q] = ngood == 0 ? 0 : -1; }
        return;
    }
    for (int g = tid; g < ngood; g += 256) { rank_all[g] = 0; rank_good[g] = 0; }
    __syncthreads();
    const int lane = tid & 63;
    for (int g = 0; g < ngood; ++g) {
        const int gi = good[g];
        const float sg = row[gi];
        int before = 0, before_good = 0;
        for (int j = tid; j < ng; j += 256) {
            const float sj = row[j];
            const bool ahead = (sj > sg) || (sj == sg && j > gi);
            if (!ahead) continue;
            const bool same = gl[j] == pid;
            const bool junk = gl[j] == -1 || (same && gc[j] == cam);
            if (junk) continue;
            ++before;
            if (same) ++before_good;  // same pid and not junk -> other camera -> good
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); before_good += __shfl_xor(before_good, o); }
        if (lane == 0) { atomicAdd(&rank_all[g], before); atomicAdd(&rank_good[g], before_good); }
    }
    __syncthreads();
    for (int g = tid; g < ngood; g += 256) slot_rank[rank_good[g]] = rank_all[g];  // rank_good is a permutation of 0..ngood-1
    __syncthreads();
    if (tid == 0) {
        double a = 0.0;
        for (int i = 0; i < ngood; ++i) {
            const int rg = slot_rank[i];
            const double d_recall = 1.0 / ngood;
            const double precision = (i + 1) * 1.0 / (rg + 1);
            const double old_precision = rg != 0 ? i * 1.0 / rg : 1.0;
            a = a + d_recall * (old_precision + precision) / 2;
        }
        ap[q] = a;
        valid[q] = 1;
        atomicAdd(&first_hist[slot_rank[0]], 1);
    }
}

// DIoU, fp64, same operation order as numpy in modification_deepsort/iou_matching.py:24-47; FMA contraction is
// disabled so every intermediate is rounded exactly as the reference does.
__global__ void diou_kernel(const double* __restrict__ tracks, int t, const double* __restrict__ dets, int m,
                            double* __restrict__ out, int as_cost) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t * m) return;
    const double* b = tracks + (long long)(i / m) * 4;
    const double* c = dets + (long long)(i % m) * 4;
    const double btlx = b[0], btly = b[1], bbrx = b[0] + b[2], bbry = b[1] + b[3];
    const double ctlx = c[0], ctly = c[1], cbrx = c[0] + c[2], cbry = c[1] + c[3];
    const double bc0 = (btly + bbry) / 2, bc1 = (btlx + bbrx) / 2;
    const double cc0 = (ctly + cbry) / 2, cc1 = (ctlx + cbrx) / 2;
    const double e0 = bc0 - cc0, e1 = bc1 - cc1;
    const double d = e0 * e0 + e1 * e1;
    const double o0 = fmin(btlx, ctlx) - fmax(bbrx, cbrx), o1 = fmin(btly, ctly) - fmax(bbry, cbry);
    const double rou = o0 * o0 + o1 * o1;
    const double w = fmax(0.0, fmin(bbrx, cbrx) - fmax(btlx, ctlx));
    const double h = fmax(0.0, fmin(bbry, cbry) - fmax(btly, ctly));
    const double inter = w * h;
    const double iou = inter / (b[2] * b[3] + c[2] * c[3] - inter);
    const double v = iou - d / rou;
    out[i] = as_cost ? 1.0 - v : v;
}

}  // namespace

int launch_argmin_rows(reid_ctx* ctx, const float* dist, int m, int n, long long ld, int32_t* idx, float* val) {
    prof_begin(ctx, REID_K_SELECT, 0, (double)m * n * 4.0);
    hipLaunchKernelGGL(argmin_rows_kernel, dim3(m), dim3(256), 0, ctx->stream, dist, n, ld, idx, val);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_topk_rows(reid_ctx* ctx, const float* dist, int m, int n, long long ld, int k, float* D, int32_t* I) {
    prof_begin(ctx, REID_K_SELECT, 0, (double)m * n * 4.0 * k);
    hipLaunchKernelGGL(topk_rows_kernel, dim3(m), dim3(256), 0, ctx->stream, dist, n, ld, k, D, I);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_rank_eval(reid_ctx* ctx, const float* score, int nq, int ng, long long ld, const long long* ql,
                     const long long* qc, const long long* gl, const long long* gc, int32_t* first_hist, double* ap,
                     int32_t* valid) {
    prof_begin(ctx, REID_K_SELECT, 0, (double)nq * ng * 4.0);
    hipLaunchKernelGGL(rank_eval_kernel, dim3(nq), dim3(256), 0, ctx->stream, score, ng, ld, ql, qc, gl, gc, first_hist, ap,
                       valid);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_diou_cost(reid_ctx* ctx, const double* tracks, int t, const double* dets, int m, double* out, int as_cost) {
    const int total = t * m;
    prof_begin(ctx, REID_K_SELECT, 0, 32.0 * (t + m) + 8.0 * total);
    hipLaunchKernelGGL(diou_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, tracks, t, dets, m, out, as_cost);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
