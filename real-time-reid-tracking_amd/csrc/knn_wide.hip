// Brute-force k-NN (reid/faiss_utils.py:56-139, METRIC_L2 = squared distances) for LARGE problems on the f16 matrix pipe, with the
// exact-fp32 answer: the re-ranking search of reid/faiss_utils.py:149-176 at N = 19 281, D = 1 263 is 0.94 TFLOP, 9.3 ms of the
// 10.8 ms re-ranking on the fp32 MFMA (101 TF/s) - the one place where that pipe, not parity, limits a row of SURVEY.md 8(f).
//
//   1. candidates: x . y^T in fp32-class arithmetic (x packed to [xh | xl'], y split as [yh 2^11 | yh | yl'], the Swin linear build
//      of gemm_f16.hip with K = 3 d virtual columns; three f16 products per multiply, fp32 accumulate) with the epilogue computing
//      |y|^2 - 2 x.y straight away (bias = |y|^2, scale = -2 / 2^11): an m x n matrix of approximate distances less the row constant;
//   2. rowsel_kernel: the KK = 32 smallest entries of every row in two sweeps of the row (threshold = the largest of the 32
//      group minima, groups = column mod 32; survivors ranked by counting);
//   3. refine_kernel: the candidates' distances again in EXACT fp32 - an fmaf chain in the MFMA loop's k order, bit for bit what the fp32 MFMA of
//      gemm_f32_dma.hip / dist_select.hip accumulates - and the k smallest of them by (distance, index): the same numbers and the
//      same order as the fused fp32 search returns;
//   4. proof per row that no non-candidate can belong to the answer: every non-candidate's approximate value is >= the 32nd
//      candidate's, so its exact value is >= that minus the error bound of step 1; if the k-th exact value does not stay below
//      that, the row is flagged and exact_row_kernel redoes it from all n columns (rare: the margin is 32 - k candidates wide).
#include "reid_internal.h"
#include <math.h>
#include <string.h>

namespace {

typedef _Float16 f16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KK = 32;        // candidates per row
constexpr int CAP = 1024;     // survivors of the threshold sweep a block can hold

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
__device__ __forceinline__ unsigned key32(float v) {
    unsigned int u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Eight terms of a dot product in the order the fp32 MFMA loops of gemm_f32_dma.hip / dist_select.hip accumulate them: a lane half
// reads four consecutive k of a 16-byte chunk, v_mfma_f32_32x32x2_f32 number e takes element e of the low half's chunk (k = e) and of
// the high half's (k = 4 + e), and the instruction itself is a two-step fmaf chain: k = 0, 4, 1, 5, 2, 6, 3, 7.
__device__ __forceinline__ float mfma_order_dot8(const float* __restrict__ a, const float* __restrict__ b, float acc) {
    const f32x4 a0 = *(const f32x4*)a, a1 = *(const f32x4*)(a + 4), b0 = *(const f32x4*)b, b1 = *(const f32x4*)(b + 4);
    acc = fmaf(a0.x, b0.x, acc); acc = fmaf(a1.x, b1.x, acc);
    acc = fmaf(a0.y, b0.y, acc); acc = fmaf(a1.y, b1.y, acc);
    acc = fmaf(a0.z, b0.z, acc); acc = fmaf(a1.z, b1.z, acc);
    acc = fmaf(a0.w, b0.w, acc); acc = fmaf(a1.w, b1.w, acc);
    return acc;
}

// ---- the KK smallest of every row (ascending by (value, column)), one block per row
__global__ __launch_bounds__(256) void rowsel_kernel(const float* __restrict__ mat, int n, long long ld, unsigned long long* __restrict__ keys,
                                                     int* __restrict__ flags) {
    __shared__ unsigned gmin[KK];
    __shared__ unsigned long long list[CAP];
    __shared__ int cnt;
    const float* row = mat + (long long)blockIdx.x * ld;
    const int tid = threadIdx.x;
    if (tid < KK) gmin[tid] = 0xffffffffu;
    if (tid == 0) cnt = 0;
    __syncthreads();
    // sweep 1: minimum of every group.  16-byte loads (the row pitch is a multiple of 64 floats): thread t takes columns 4 t .. 4 t + 3
    // of every 1024, whose groups (column mod 32) are 4 (t mod 8) + e - the same four for all of a thread's columns
    unsigned m4[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    const int n4 = n & ~3;
    for (int j = tid * 4; j < n4; j += 1024) {
        const f32x4 v = *(const f32x4*)(row + j);
        const unsigned k0 = key32(v.x), k1 = key32(v.y), k2 = key32(v.z), k3 = key32(v.w);
        m4[0] = k0 < m4[0] ? k0 : m4[0]; m4[1] = k1 < m4[1] ? k1 : m4[1];
        m4[2] = k2 < m4[2] ? k2 : m4[2]; m4[3] = k3 < m4[3] ? k3 : m4[3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicMin(&gmin[(tid * 4 + e) & (KK - 1)], m4[e]);
    if (tid < n - n4) atomicMin(&gmin[(n4 + tid) & (KK - 1)], key32(row[n4 + tid]));
    __syncthreads();
    unsigned thr = 0;
#pragma unroll
    for (int g = 0; g < KK; ++g) thr = gmin[g] > thr ? gmin[g] : thr;     // at least KK entries lie at or below it
    // sweep 2 (the row comes from L2 now): survivors
    for (int j = tid * 4; j < n4; j += 1024) {
        const f32x4 v = *(const f32x4*)(row + j);
        const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (key32(ve[e]) <= thr) {
                const int s = atomicAdd(&cnt, 1);
                if (s < CAP) list[s] = pack_key(ve[e], j + e);
            }
        }
    }
    if (tid < n - n4 && key32(row[n4 + tid]) <= thr) {
        const int s = atomicAdd(&cnt, 1);
        if (s < CAP) list[s] = pack_key(row[n4 + tid], n4 + tid);
    }
    __syncthreads();
    const int c = cnt < CAP ? cnt : CAP;
    if (cnt > CAP && tid == 0) flags[blockIdx.x] = 1;     // (thousands of ties at the threshold: the exact path takes the row)
    unsigned long long* out = keys + (long long)blockIdx.x * KK;
    for (int e = tid; e < c; e += 256) {
        const unsigned long long me = list[e];
        int rank = 0;
        for (int o = 0; o < c; ++o) rank += list[o] < me;
        if (rank < KK) out[rank] = me;
    }
    if (c < KK)
        for (int e = c + tid; e < KK; e += 256) out[e] = ~0ull;
}

// ---- exact distances of the candidates, the k smallest of them, and the proof that the candidates sufficed.
// 64 lanes = two rows x 32 candidates.
__global__ __launch_bounds__(256) void refine_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ y, long long ldy, int K,
                                                     const float* __restrict__ rs, const float* __restrict__ cq, const float* __restrict__ sc,
                                                     const unsigned long long* __restrict__ keys, int rows, int n, int k, float err_scale,
                                                     float* __restrict__ D, int32_t* __restrict__ I, int* __restrict__ flags) {
    const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;
    const bool live = row < rows;
    const unsigned long long ak = live ? keys[(long long)row * KK + c] : ~0ull;
    const int idx = (int)(ak & 0xffffffffu);
    const bool valid = live && ak != ~0ull && idx < n;
    float dot = 0.f;
    if (valid) {
        const float* xr = x + (long long)row * ldx;
        const float* yr = y + (long long)idx * ldy;
        for (int q = 0; q < K; q += 8) dot = mfma_order_dot8(xr + q, yr + q, dot);   // K % 64 == 0 (rows are zero-padded)
    }
    const float r2 = live ? rs[row] : 0.f;
    const float e = valid ? l2sqr_of(dot, r2, cq[idx]) : INFINITY;
    const unsigned long long ek = valid ? pack_key(e, idx) : ~0ull;
    // rank among the 32 candidates of this row (keys are unique: they carry the column)
    int rank = 0;
#pragma unroll
    for (int o = 0; o < 32; ++o) {
        const unsigned long long other = __shfl(ek, half * 32 + o);
        rank += other < ek;
    }
    if (live && rank < k) {
        D[(long long)row * k + rank] = valid ? e : INFINITY;
        I[(long long)row * k + rank] = valid ? idx : -1;
    }
    // proof: the k-th exact value, in the matrix's units (without |x|^2), stays below the 32nd approximate value minus the error
    // of the approximate arithmetic
    const int kth = k < 32 ? k - 1 : 31;
    const float ss = sc[2];                                        // the candidate matrix is in units of sx sy
    float ek_val = valid && rank == kth ? (e - r2) * ss : -INFINITY;   // exactly one lane of the row holds rank kth (if it is valid)
    float a_last = c == 31 ? (ak != ~0ull ? unpack_val(ak) : INFINITY) : -INFINITY;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        ek_val = fmaxf(ek_val, __shfl_xor(ek_val, o));
        a_last = fmaxf(a_last, __shfl_xor(a_last, o));
    }
    if (live && c == 0) {
        const float err = err_scale * (r2 + sc[3]) * ss;
        const bool enough = n <= KK;                               // every column is a candidate
        if (!enough && !(ek_val + err < a_last - err)) flags[row] = 1;
    }
}

// ---- a flagged row from scratch: exact distances to all n columns into the row's slot of the matrix, then k rounds of
// "smallest key above the last one"
__global__ __launch_bounds__(256) void exact_row_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ y, long long ldy,
                                                        int K, const float* __restrict__ rs, const float* __restrict__ cq, float* __restrict__ mat,
                                                        long long ld, int n, int k, float* __restrict__ D, int32_t* __restrict__ I,
                                                        const int* __restrict__ flags, int* __restrict__ nflagged) {
    const int row = blockIdx.x;
    if (!flags[row]) return;
    __shared__ unsigned long long sh[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) atomicAdd(nflagged, 1);
    float* out = mat + (long long)row * ld;
    const float* xr = x + (long long)row * ldx;
    const float r2 = rs[row];
    for (int j = tid; j < n; j += 256) {
        const float* yr = y + (long long)j * ldy;
        float dot = 0.f;
        for (int q = 0; q < K; q += 8) dot = mfma_order_dot8(xr + q, yr + q, dot);
        out[j] = l2sqr_of(dot, r2, cq[j]);
    }
    __syncthreads();
    unsigned long long last = 0;
    for (int r = 0; r < k; ++r) {
        unsigned long long best = ~0ull;
        for (int j = tid; j < n; j += 256) {
            const unsigned long long key = pack_key(out[j], j);
            if ((r == 0 || key > last) && key < best) best = key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o);
            best = other < best ? other : best;
        }
        __syncthreads();
        if (lane == 0) sh[wave] = best;
        __syncthreads();
        best = sh[0];
        for (int w = 1; w < 4; ++w) best = sh[w] < best ? sh[w] : best;
        if (tid == 0) {
            D[(long long)row * k + r] = best == ~0ull ? INFINITY : unpack_val(best);
            I[(long long)row * k + r] = best == ~0ull ? -1 : (int32_t)(best & 0xffffffffu);
        }
        last = best;
    }
}

// The candidate stage works on NORMALISED operands (the k-NN entry point takes any features; the split needs |value| 2^11 inside
// f16): sc[0] = sx, sc[1] = sy powers of two with max |x| sx <= 1 and max |y| sy <= 1, sc[2] = sx sy, sc[3] = max |y|^2.  Scaling by
// a power of two is exact and a positive factor keeps every row's order, so the candidate matrix holds sx sy (|y|^2 - 2 x.y).
__global__ __launch_bounds__(256) void scale_prep_kernel(const float* __restrict__ rs, int nq, const float* __restrict__ cq, int nb,
                                                         float* __restrict__ sc) {
    __shared__ float sh[2][4];
    float mx = 0.f, my = 0.f;
    for (int i = threadIdx.x; i < nq; i += 256) mx = fmaxf(mx, rs[i]);
    for (int i = threadIdx.x; i < nb; i += 256) my = fmaxf(my, cq[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        my = fmaxf(my, __shfl_xor(my, o));
    }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = mx; sh[1][threadIdx.x >> 6] = my; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(sh[0][0], sh[0][1]), fmaxf(sh[0][2], sh[0][3]));
        my = fmaxf(fmaxf(sh[1][0], sh[1][1]), fmaxf(sh[1][2], sh[1][3]));
        auto pow2_inv = [](float sq) {      // 2^-e with 2^e >= sqrt(sq); 1 for an all-zero (or non-finite) operand
            if (!(sq > 0.f) || !(sq < INFINITY)) return 1.0f;
            int e;
            (void)frexpf(sqrtf(sq), &e);    // sqrt(sq) = f 2^e, f in [0.5, 1)
            e = e < -100 ? -100 : (e > 100 ? 100 : e);
            return ldexpf(1.0f, -e);
        };
        const float sx = pow2_inv(mx), sy = pow2_inv(my);
        sc[0] = sx; sc[1] = sy; sc[2] = sx * sy; sc[3] = my;
    }
}
__global__ __launch_bounds__(256) void scaled_bias_kernel(const float* __restrict__ cq, int nb, const float* __restrict__ sc, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nb) out[i] = cq[i] * sc[2];
}

}  // namespace

bool knn_wide_eligible(reid_ctx* ctx, int nq, int nb, int d, int k) {
    return ctx->knn_wide && k >= 1 && k <= 24 && nb >= 4096 && d >= 128 && (long long)nq * nb >= ctx->knn_wide_min;
}

// xp [nq][ld] / yp [nb][ld]: the zero-padded operands of the fused fp32 search (ld % 64 == 0), rs / cq their squared norms.
// force_flags != 0 (tests): every row whose index is a multiple of it takes the exact fallback.
int knn_wide_dev(reid_ctx* ctx, const float* xp, int nq, const float* yp, int nb, int ld, const float* rs, const float* cq, int k, float* d_D,
                 int32_t* d_I) {
    ARG_CHECK(ld % 64 == 0 && k <= 24);
    const int nbpad = (nb + 255) / 256 * 256;       // whole 256-column GEMM tiles; the matrix rows have this pitch (16-byte aligned)
    f16 *a16, *w16;
    float *mat, *sc, *biasv;
    unsigned long long* keys;
    int *flags, *nflag;
    REID_TRY(ctx_ws(ctx, "knnw.a16", (size_t)nq * 2 * ld * 2, (void**)&a16));
    REID_TRY(ctx_ws(ctx, "knnw.w16", (size_t)nbpad * 3 * ld * 2, (void**)&w16));
    REID_TRY(ctx_ws(ctx, "knnw.sc", 16, (void**)&sc));
    REID_TRY(ctx_ws(ctx, "knnw.bias", (size_t)nb * 4, (void**)&biasv));
    REID_TRY(ctx_ws(ctx, "knnw.flags", (size_t)nq * 4 + 16, (void**)&flags));
    nflag = flags + nq;
    HIP_TRY(hipMemsetAsync(flags, 0, (size_t)nq * 4 + 16, ctx->stream));
    if (ctx->knn_wide_force > 0) {      // tests: exercise the exact fallback on some rows
        std::vector<int> h(nq, 0);
        for (int i = 0; i < nq; i += ctx->knn_wide_force) h[i] = 1;
        HIP_TRY(hipMemcpyAsync(flags, h.data(), (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    hipLaunchKernelGGL(scale_prep_kernel, dim3(1), dim3(256), 0, ctx->stream, rs, nq, cq, nb, sc);
    hipLaunchKernelGGL(scaled_bias_kernel, dim3((nb + 255) / 256), dim3(256), 0, ctx->stream, cq, nb, sc, biasv);
    LAUNCH_CHECK();
    REID_TRY(launch_split_pack(ctx, xp, nq, ld, a16, sc));
    if (nbpad > nb) HIP_TRY(hipMemsetAsync(w16 + (size_t)nb * 3 * ld, 0, (size_t)(nbpad - nb) * 3 * ld * 2, ctx->stream));
    REID_TRY(launch_split_weights(ctx, yp, nb, 1, ld, 3, w16, sc + 1));
    // query tiles bound the scratch matrix to 1 GiB; whole 256-row GEMM tiles
    long long rows_per = ((long long)1 << 28) / nbpad / 256 * 256;
    if (rows_per < 256) rows_per = 256;
    const int tile = nq < rows_per ? nq : (int)rows_per;
    REID_TRY(ctx_ws(ctx, "knnw.mat", (size_t)tile * nbpad * 4, (void**)&mat));
    REID_TRY(ctx_ws(ctx, "knnw.keys", (size_t)tile * KK * 8, (void**)&keys));
    // |approximate - exact| of |y|^2 - 2 x.y, worst case: the split operands carry 22 bits and the xl.yl term is dropped (3 x 2^-22
    // |x||y|), and the fp32 accumulator is rounded once or twice per MFMA over a chain of 3 ld / 16 of them (each at most 2^-24 of the
    // running sum <= 2^11 sum |x_k y_k|); times two for the factor -2, and |x||y| <= (|x|^2 + |y|^2) / 2
    const float err_scale = (6.0f * (float)ld / 16.0f + 16.0f) / 16777216.0f;
    for (int i = 0; i < nq; i += tile) {
        const int m = nq - i < tile ? nq - i : tile;
        Gemm16Params q;
        memset(&q, 0, sizeof(q));
        q.A = a16 + (size_t)i * 2 * ld; q.lda = 2 * ld;
        q.B = w16; q.ldb = 3 * ld;
        q.M = m; q.N = nbpad; q.K = 3 * ld;
        q.C32 = mat; q.ldc = nbpad;
        q.col_shift = biasv; q.lin = 1; q.n_real = nb;
        q.split_terms = 3; q.a_k = 2 * ld; q.acc_scale = -2.0f / 2048.0f;
        q.zero_page = ctx->se18.zero_page;
        // a long K loop (3 ld / 64 tiles): the 256-wide tile with 64-deep K-tiles (4.7 against 5.2 ms at 19 281 x 19 281 x 1 263; the
        // Swin linears, whose launches are their epilogues, keep the 128-wide build)
        const int cfg0 = ctx->f16_cfg;
        if (!cfg0) ctx->f16_cfg = 256642;
        const int st = launch_gemm_f16(ctx, A16_DENSE, q, REID_K_DIST_GEMM, 2.0 * m * nb * ld, 4.0 * ((double)m * ld + (double)nb * ld + (double)m * nb));
        ctx->f16_cfg = cfg0;
        REID_TRY(st);
        prof_begin(ctx, REID_K_SELECT, 0, (double)m * nb * 8.0);
        hipLaunchKernelGGL(rowsel_kernel, dim3(m), dim3(256), 0, ctx->stream, mat, nb, (long long)nbpad, keys, flags + i);
        hipLaunchKernelGGL(refine_kernel, dim3((m + 7) / 8), dim3(256), 0, ctx->stream, xp + (size_t)i * ld, (long long)ld, yp, (long long)ld, ld,
                           rs + i, cq, sc, keys, m, nb, k, err_scale, d_D + (size_t)i * k, d_I + (size_t)i * k, flags + i);
        hipLaunchKernelGGL(exact_row_kernel, dim3(m), dim3(256), 0, ctx->stream, xp + (size_t)i * ld, (long long)ld, yp, (long long)ld, ld, rs + i, cq,
                           mat, (long long)nbpad, nb, k, d_D + (size_t)i * k, d_I + (size_t)i * k, flags + i, nflag);
        prof_end(ctx);
        LAUNCH_CHECK();
    }
    return REID_OK;
}
