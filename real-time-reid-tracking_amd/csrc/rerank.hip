// k-reciprocal Jaccard re-ranking on the device: the reference's `compute_jaccard_distance`
// (/root/reference/reid/faiss_utils.py:140-244), which after one GPU k-NN runs O(N) Python loops over dense N x N numpy
// matrices (N = 19 281 on Market: 1.5 GB each, minutes of CPU time).
//
// MI355X layout: V is never dense.  A row of V has at most W1 = k1 + k1*(round(k1/2)+1) non-zeros (240 for k1 = 20), after
// local query expansion at most k2*W1 (1 440): both live in HBM as ELL rows (index, value), the transposed V as CSC lists.
// The only dense object is the N x N answer itself, written once, coalesced.  One Jaccard row is accumulated in LDS
// (N floats, 77 KB at N = 19 281; a per-block HBM scratch row takes over when N floats do not fit in 160 KB):
//     S[i][j] = sum_c min(V[i][c], V[j][c])  =  for c in nz(V[i]): for (j, v) in column c: S[j] += min(V[i][c], v)
// i.e. O(nnz_row * nnz_col) LDS atomics per row instead of the reference's N-wide numpy temporaries.
//
// Integer steps (reciprocal sets, expansion rule, unique) are bit-exact with the reference; the float steps use the same
// fp32 formulas, with sums whose order differs (softmax denominator, LDS-atomic min-sums): tolerance, not bit-exact.
#include "reid_internal.h"

namespace {

constexpr int MAX_K1 = 64;

// ---- 1. k-reciprocal neighbour lists R(i, k1) and R(i, kh)   (k_reciprocal_neigh, faiss_utils.py:140-144) -------------
// One wave per i; lane a owns forward neighbour rank[i][a].  kh1 = min(kh + 1, k1) columns for the half lists.
__global__ __launch_bounds__(256) void recip_kernel(const int32_t* __restrict__ rank, int n, int k1, int kh1,
                                                    int32_t* __restrict__ r1_idx, int32_t* __restrict__ r1_cnt,
                                                    int32_t* __restrict__ rh_idx, int32_t* __restrict__ rh_cnt) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    bool in1 = false, inh = false;
    int c = -1;
    if (lane < k1) {
        c = rank[(long long)i * k1 + lane];
        if (c >= 0 && c < n) {
            const int32_t* back = rank + (long long)c * k1;
            for (int b = 0; b < k1; ++b) {
                const bool hit = back[b] == i;
                in1 |= hit;
                inh |= hit && b < kh1;
            }
            inh = inh && lane < kh1;
        }
    }
    const unsigned long long m1 = __ballot(in1), mh = __ballot(inh);
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (in1) r1_idx[(long long)i * k1 + __popcll(m1 & below)] = c;
    if (inh) rh_idx[(long long)i * kh1 + __popcll(mh & below)] = c;
    if (lane == 0) {
        r1_cnt[i] = __popcll(m1);
        rh_cnt[i] = __popcll(mh);
    }
}

// ---- 2. expansion set + softmax weights   (faiss_utils.py:190-204) ---------------------------------------------------
// One block per i.  LDS: x_i [d] | cand [w1] | sorted [w1] | keep [w1] | logit [w1]
__global__ __launch_bounds__(256) void expand_kernel(const float* __restrict__ x, int n, int d, int k1, int kh1, int w1,
                                                     const int32_t* __restrict__ r1_idx, const int32_t* __restrict__ r1_cnt,
                                                     const int32_t* __restrict__ rh_idx, const int32_t* __restrict__ rh_cnt,
                                                     int32_t* __restrict__ v_idx, float* __restrict__ v_val,
                                                     int32_t* __restrict__ v_cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xi = (float*)smem;
    int* cand = (int*)(xi + d);
    int* sorted = cand + w1;
    int* keep = sorted + w1;
    float* logit = (float*)(keep + w1);
    __shared__ int R[MAX_K1], off[MAX_K1], flag[MAX_K1];
    __shared__ int total_sh, nuniq_sh;
    __shared__ float red[8];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nr = r1_cnt[i];
    for (int k = tid; k < d; k += 256) xi[k] = x[(long long)i * d + k];
    if (tid < nr) R[tid] = r1_idx[(long long)i * k1 + tid];
    if (tid == 0) nuniq_sh = 0;
    __syncthreads();
    // candidate a joins when more than 2/3 of its half-list lies inside R (same double comparison as the reference)
    for (int a = wave; a < nr; a += 4) {
        const int c = R[a];
        const int len = rh_cnt[c];
        bool in = false;
        if (lane < len) {
            const int e = rh_idx[(long long)c * kh1 + lane];
            for (int t = 0; t < nr; ++t) in |= R[t] == e;
        }
        const int common = __popcll(__ballot(in));
        if (lane == 0) flag[a] = (double)common > (2.0 / 3.0) * (double)len ? len : 0;
    }
    __syncthreads();
    if (tid == 0) {
        int base = nr;
        for (int a = 0; a < nr; ++a) {
            off[a] = base;
            base += flag[a];
        }
        total_sh = base;
    }
    __syncthreads();
    const int total = total_sh;
    if (tid < nr) cand[tid] = R[tid];
    for (int a = wave; a < nr; a += 4)
        if (flag[a] && lane < flag[a]) cand[off[a] + lane] = rh_idx[(long long)R[a] * kh1 + lane];
    __syncthreads();
    // np.unique: keep the first occurrence, position = number of kept smaller values
    for (int p = tid; p < total; p += 256) {
        const int v = cand[p];
        bool first = true;
        for (int q = 0; q < p; ++q) first &= cand[q] != v;
        keep[p] = first;
    }
    __syncthreads();
    for (int p = tid; p < total; p += 256) {
        if (!keep[p]) continue;
        const int v = cand[p];
        int pos = 0;
        for (int q = 0; q < total; ++q) pos += keep[q] && cand[q] < v;
        sorted[pos] = v;
        atomicAdd(&nuniq_sh, 1);
    }
    __syncthreads();
    const int nu = nuniq_sh;
    // dist = 2 - 2 x_i . x_e ; V[i][e] = softmax(-dist)
    for (int e = wave; e < nu; e += 4) {
        const float* xe = x + (long long)sorted[e] * d;
        float s = 0.f;
        for (int k = lane; k < d; k += 64) s += xi[k] * xe[k];
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) logit[e] = -(2.f - 2.f * s);
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int e = tid; e < nu; e += 256) mx = fmaxf(mx, logit[e]);
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int e = tid; e < nu; e += 256) {
        const float w = expf(logit[e] - mx);
        logit[e] = w;
        sum += w;
    }
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    for (int e = tid; e < nu; e += 256) {
        v_idx[(long long)i * w1 + e] = sorted[e];
        v_val[(long long)i * w1 + e] = logit[e] / sum;
    }
    if (tid == 0) v_cnt[i] = nu;
}

// ---- 3. local query expansion  V_qe[i] = mean(V[rank[i][:k2]])   (faiss_utils.py:208-213) --------------------------
// Persistent blocks; `acc` is a dense row of N floats (LDS, or this block's HBM scratch row), all zero between rows.
// The k2 source rows are added one after the other (the reference's np.mean order); inside one row indices are unique.
template <bool LDS_ACC>
__global__ __launch_bounds__(256) void qe_kernel(const int32_t* __restrict__ rank, int n, int k1, int k2, int w1, int w2,
                                                 const int32_t* __restrict__ v_idx, const float* __restrict__ v_val,
                                                 const int32_t* __restrict__ v_cnt, float* __restrict__ scratch,
                                                 int32_t* __restrict__ q_idx, float* __restrict__ q_val,
                                                 int32_t* __restrict__ q_cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* touched = (int*)smem;                                   // [w2]
    float* acc = LDS_ACC ? (float*)(touched + w2) : scratch + (long long)blockIdx.x * n;
    __shared__ int nout;
    const int tid = threadIdx.x;
    if (LDS_ACC)
        for (int j = tid; j < n; j += 256) acc[j] = 0.f;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        if (tid == 0) nout = 0;
        __syncthreads();
        for (int r = 0; r < k2; ++r) {
            const int row = rank[(long long)i * k1 + r];
            if (row >= 0 && row < n) {
                const int cnt = v_cnt[row];
                for (int t = tid; t < cnt; t += 256) {
                    const int idx = v_idx[(long long)row * w1 + t];
                    const float old = acc[idx];
                    if (old == 0.f) touched[atomicAdd(&nout, 1)] = idx;
                    acc[idx] = old + v_val[(long long)row * w1 + t];
                }
            }
            __syncthreads();
        }
        const int cnt = nout;
        for (int t = tid; t < cnt; t += 256) {
            const int idx = touched[t];
            q_idx[(long long)i * w2 + t] = idx;
            q_val[(long long)i * w2 + t] = acc[idx] / (float)k2;
            acc[idx] = 0.f;
        }
        if (tid == 0) q_cnt[i] = cnt;
        __syncthreads();
    }
}

// ---- 4. transpose of V (column lists): count, scan, fill   (invIndex, faiss_utils.py:217-219) ----------------------
__global__ void col_count_kernel(const int32_t* __restrict__ q_idx, const int32_t* __restrict__ q_cnt, int n, int w,
                                 int32_t* __restrict__ col_cnt) {
    const int i = blockIdx.x;
    const int cnt = q_cnt[i];
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) atomicAdd(&col_cnt[q_idx[(long long)i * w + t]], 1);
}

__global__ __launch_bounds__(1024) void col_scan_kernel(const int32_t* __restrict__ col_cnt, int n,
                                                        long long* __restrict__ col_start) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < n ? lo + per : n;
    long long s = 0;
    for (int j = lo; j < hi; ++j) s += col_cnt[j];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int t = 0; t < 1024; ++t) {
            const long long v = part[t];
            part[t] = run;
            run += v;
        }
        col_start[n] = run;
    }
    __syncthreads();
    s = part[tid];
    for (int j = lo; j < hi; ++j) {
        col_start[j] = s;
        s += col_cnt[j];
    }
}

__global__ void col_fill_kernel(const int32_t* __restrict__ q_idx, const float* __restrict__ q_val,
                                const int32_t* __restrict__ q_cnt, int n, int w, const long long* __restrict__ col_start,
                                int32_t* __restrict__ col_cur, int32_t* __restrict__ col_row, float* __restrict__ col_val) {
    const int i = blockIdx.x;
    const int cnt = q_cnt[i];
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
        const int c = q_idx[(long long)i * w + t];
        const long long pos = col_start[c] + atomicAdd(&col_cur[c], 1);
        col_row[pos] = i;
        col_val[pos] = q_val[(long long)i * w + t];
    }
}

// ---- 5. Jaccard rows   (faiss_utils.py:221-237) ---------------------------------------------------------------------
// out[i][j] = max(0, 1 - S/(2 - S)),  S = sum_c min(V[i][c], V[j][c]).  One wave per non-zero of row i walks its column.
template <bool LDS_ACC>
__global__ __launch_bounds__(512) void jaccard_kernel(int n, int w, const int32_t* __restrict__ q_idx,
                                                      const float* __restrict__ q_val, const int32_t* __restrict__ q_cnt,
                                                      const long long* __restrict__ col_start,
                                                      const int32_t* __restrict__ col_row, const float* __restrict__ col_val,
                                                      float* __restrict__ scratch, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc = LDS_ACC ? (float*)smem : scratch + (long long)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (LDS_ACC)
        for (int j = tid; j < n; j += 512) acc[j] = 0.f;
    __syncthreads();
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int cnt = q_cnt[i];
        for (int e = wave; e < cnt; e += 8) {
            const int c = q_idx[(long long)i * w + e];
            const float v = q_val[(long long)i * w + e];
            const long long lo = col_start[c], hi = col_start[c + 1];
            for (long long t = lo + lane; t < hi; t += 64) atomicAdd(&acc[col_row[t]], fminf(v, col_val[t]));
        }
        __syncthreads();
        float* o = out + (size_t)i * n;
        for (int j = tid; j < n; j += 512) {
            const float s = acc[j];
            acc[j] = 0.f;
            const float jac = 1.f - s / (2.f - s);
            o[j] = jac < 0.f ? 0.f : jac;
        }
        __syncthreads();
    }
}

}  // namespace

// x: [n][d] L2-normalised rows (the reference's `dist = 2 - 2 x.y` assumes it); d_rank: [n][k1] int32 neighbour lists or
// nullptr (then the library's own brute-force squared-L2 k-NN, self included, supplies them); d_out: [n][n].
extern "C" int reid_rerank_jaccard_dev(reid_ctx* ctx, const float* d_x, int n, int d, int k1, int k2, const int32_t* d_rank,
                                       float* d_out) {
    ARG_CHECK(ctx && d_x && d_out && n >= 1 && d >= 1 && k1 >= 1 && k1 <= MAX_K1 && k1 <= n && k2 >= 1);
    CTX_GUARD(ctx);
    const int kh = (int)nearbyint(k1 / 2.0);   // np.around: half to even
    const int kh1 = kh + 1 < k1 ? kh + 1 : k1;
    const int k2e = k2 < k1 ? k2 : k1;         // initial_rank[i, :k2] has at most k1 columns
    const int w1 = k1 + k1 * kh1;
    const long long w2l = k2e == 1 ? w1 : (long long)k2e * w1;
    const int w2 = (int)(w2l < n ? w2l : n);
    hipStream_t st = ctx->stream;

    int32_t* rank = nullptr;
    if (d_rank) {
        rank = const_cast<int32_t*>(d_rank);
    } else {
        float* knn_d;
        REID_TRY(ctx_ws(ctx, "rr.knnD", (size_t)n * k1 * 4, (void**)&knn_d));
        REID_TRY(ctx_ws(ctx, "rr.rank", (size_t)n * k1 * 4, (void**)&rank));
        REID_TRY(reid_knn_dev(ctx, d_x, n, d_x, n, d, k1, knn_d, rank));
    }
    int32_t *r1_idx, *r1_cnt, *rh_idx, *rh_cnt, *v_idx, *v_cnt, *q_idx, *q_cnt, *col_cnt, *col_cur, *col_row;
    float *v_val, *q_val, *col_val;
    long long* col_start;
    REID_TRY(ctx_ws(ctx, "rr.r1i", (size_t)n * k1 * 4, (void**)&r1_idx));
    REID_TRY(ctx_ws(ctx, "rr.r1c", (size_t)n * 4, (void**)&r1_cnt));
    REID_TRY(ctx_ws(ctx, "rr.rhi", (size_t)n * kh1 * 4, (void**)&rh_idx));
    REID_TRY(ctx_ws(ctx, "rr.rhc", (size_t)n * 4, (void**)&rh_cnt));
    REID_TRY(ctx_ws(ctx, "rr.vi", (size_t)n * w1 * 4, (void**)&v_idx));
    REID_TRY(ctx_ws(ctx, "rr.vv", (size_t)n * w1 * 4, (void**)&v_val));
    REID_TRY(ctx_ws(ctx, "rr.vc", (size_t)n * 4, (void**)&v_cnt));
    REID_TRY(ctx_ws(ctx, "rr.cc", (size_t)n * 4, (void**)&col_cnt));
    REID_TRY(ctx_ws(ctx, "rr.cu", (size_t)n * 4, (void**)&col_cur));
    REID_TRY(ctx_ws(ctx, "rr.cs", (size_t)(n + 1) * 8, (void**)&col_start));

    prof_begin(ctx, REID_K_SELECT, 0, 0);
    hipLaunchKernelGGL(recip_kernel, dim3((n + 3) / 4), dim3(256), 0, st, rank, n, k1, kh1, r1_idx, r1_cnt, rh_idx, rh_cnt);
    LAUNCH_CHECK();
    const size_t sh2 = (size_t)d * 4 + (size_t)w1 * 16;
    ARG_CHECK(sh2 <= 150 * 1024);
    if (sh2 > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)expand_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2));
    hipLaunchKernelGGL(expand_kernel, dim3(n), dim3(256), sh2, st, d_x, n, d, k1, kh1, w1, r1_idx, r1_cnt, rh_idx, rh_cnt,
                       v_idx, v_val, v_cnt);
    LAUNCH_CHECK();

    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const int grid = n < 2 * cus ? n : 2 * cus;
    float* scratch = nullptr;
    const size_t acc_b = (size_t)n * 4;
    if (k2e != 1) {
        REID_TRY(ctx_ws(ctx, "rr.qi", (size_t)n * w2 * 4, (void**)&q_idx));
        REID_TRY(ctx_ws(ctx, "rr.qv", (size_t)n * w2 * 4, (void**)&q_val));
        REID_TRY(ctx_ws(ctx, "rr.qc", (size_t)n * 4, (void**)&q_cnt));
        const size_t list_b = (size_t)w2 * 4;
        if (list_b + acc_b <= 150 * 1024) {
            const size_t sh = list_b + acc_b;
            if (sh > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute((const void*)qe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
            hipLaunchKernelGGL(qe_kernel<true>, dim3(grid), dim3(256), sh, st, rank, n, k1, k2e, w1, w2, v_idx, v_val, v_cnt,
                               nullptr, q_idx, q_val, q_cnt);
        } else {
            ARG_CHECK(list_b <= 150 * 1024);
            REID_TRY(ctx_ws(ctx, "rr.scratch", (size_t)grid * acc_b, (void**)&scratch));
            HIP_TRY(hipMemsetAsync(scratch, 0, (size_t)grid * acc_b, st));
            if (list_b > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute((const void*)qe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)list_b));
            hipLaunchKernelGGL(qe_kernel<false>, dim3(grid), dim3(256), list_b, st, rank, n, k1, k2e, w1, w2, v_idx, v_val,
                               v_cnt, scratch, q_idx, q_val, q_cnt);
        }
        LAUNCH_CHECK();
    } else {
        q_idx = v_idx;
        q_val = v_val;
        q_cnt = v_cnt;
    }
    const int w = k2e != 1 ? w2 : w1;
    REID_TRY(ctx_ws(ctx, "rr.cr", (size_t)n * w * 4, (void**)&col_row));
    REID_TRY(ctx_ws(ctx, "rr.cv", (size_t)n * w * 4, (void**)&col_val));
    HIP_TRY(hipMemsetAsync(col_cnt, 0, (size_t)n * 4, st));
    HIP_TRY(hipMemsetAsync(col_cur, 0, (size_t)n * 4, st));
    hipLaunchKernelGGL(col_count_kernel, dim3(n), dim3(256), 0, st, q_idx, q_cnt, n, w, col_cnt);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(col_scan_kernel, dim3(1), dim3(1024), 0, st, col_cnt, n, col_start);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(col_fill_kernel, dim3(n), dim3(256), 0, st, q_idx, q_val, q_cnt, n, w, col_start, col_cur, col_row,
                       col_val);
    LAUNCH_CHECK();
    if (acc_b <= 150 * 1024) {
        if (acc_b > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void*)jaccard_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_b));
        hipLaunchKernelGGL(jaccard_kernel<true>, dim3(grid), dim3(512), acc_b, st, n, w, q_idx, q_val, q_cnt, col_start, col_row,
                           col_val, nullptr, d_out);
    } else {
        if (!scratch) {
            REID_TRY(ctx_ws(ctx, "rr.scratch", (size_t)grid * acc_b, (void**)&scratch));
            HIP_TRY(hipMemsetAsync(scratch, 0, (size_t)grid * acc_b, st));
        }
        hipLaunchKernelGGL(jaccard_kernel<false>, dim3(grid), dim3(512), 0, st, n, w, q_idx, q_val, q_cnt, col_start, col_row,
                           col_val, scratch, d_out);
    }
    LAUNCH_CHECK();
    prof_end(ctx);
    return REID_OK;
}

extern "C" int reid_rerank_jaccard(reid_ctx* ctx, const float* x, int n, int d, int k1, int k2, const int32_t* rank,
                                   float* out) {
    ARG_CHECK(ctx && x && out && n >= 1 && d >= 1);
    CTX_GUARD(ctx);
    float *dx, *dout;
    int32_t* drank = nullptr;
    REID_TRY(ctx_ws(ctx, "rr.x", (size_t)n * d * 4, (void**)&dx));
    REID_TRY(ctx_ws(ctx, "rr.out", (size_t)n * n * 4, (void**)&dout));
    HIP_TRY(hipMemcpyAsync(dx, x, (size_t)n * d * 4, hipMemcpyHostToDevice, ctx->stream));
    if (rank) {
        ARG_CHECK(k1 >= 1);
        REID_TRY(ctx_ws(ctx, "rr.rank_in", (size_t)n * k1 * 4, (void**)&drank));
        HIP_TRY(hipMemcpyAsync(drank, rank, (size_t)n * k1 * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    REID_TRY(reid_rerank_jaccard_dev(ctx, dx, n, d, k1, k2, drank, dout));
    HIP_TRY(hipMemcpyAsync(out, dout, (size_t)n * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}
