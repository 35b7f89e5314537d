// Post-processing of the reference's evaluation script on the device:
//   * flip test-time augmentation + retrieval descriptor, image_reid_inference.py:112-123,252-253,267-268:
//         d(x) = cat(normalize(emb(x)), normalize(logits(x)));  descriptor = normalize((d(x) + d(hflip(x))) / 2)
//   * camera de-biasing, inference_utils.py:5-15 (`diminish_camera_bias`): per camera c with rows X_c,
//         P = inverse(X_c^T X_c + n_c * la * I);   X_c <- normalize_rows((X_c - mean(X_c)) P^T)
//     The reference inverts a 1263 x 1263 matrix per camera with LAPACK on the CPU.  Here everything stays a GEMM on the
//     fp32 MFMA kernel: the Gram matrix, then Newton-Schulz  X <- X (2I - A X)  from X0 = I / ||A||_inf (A is symmetric
//     positive definite with eigenvalues in [n la, ||A||_inf], so the iteration converges quadratically from the first step
//     and every iterate is a polynomial in A, i.e. symmetric: no transposes), then the projection GEMM.
#include "reid_internal.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

__global__ void flip_w_nchw_kernel(const float* __restrict__ x, long long rows, int w, float* __restrict__ y) {
    const long long total = rows * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / w;
        const int c = (int)(i - r * w);
        y[i] = x[r * w + (w - 1 - c)];
    }
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// one block (256 threads) per row: F.normalize semantics (x / max(||x||, 1e-12))
__global__ __launch_bounds__(256) void descriptor_kernel(const float* __restrict__ e1, const float* __restrict__ l1,
                                                         const float* __restrict__ e2, const float* __restrict__ l2, int de,
                                                         int dl, float* __restrict__ out) {
    __shared__ float sh[4];
    const int r = blockIdx.x, tid = threadIdx.x, d = de + dl;
    auto sq = [&](const float* p, int n) {
        float a = 0.f;
        for (int k = tid; k < n; k += 256) a += p[k] * p[k];
        return fmaxf(sqrtf(block_sum(a, sh)), 1e-12f);
    };
    const float ne1 = sq(e1 + (long long)r * de, de), nl1 = sq(l1 + (long long)r * dl, dl);
    float ne2 = 1.f, nl2 = 1.f;
    if (e2) {
        ne2 = sq(e2 + (long long)r * de, de);
        nl2 = sq(l2 + (long long)r * dl, dl);
    }
    float* o = out + (long long)r * d;
    float acc = 0.f;
    for (int k = tid; k < d; k += 256) {
        float v = k < de ? e1[(long long)r * de + k] / ne1 : l1[(long long)r * dl + k - de] / nl1;
        if (e2) {
            const float u = k < de ? e2[(long long)r * de + k] / ne2 : l2[(long long)r * dl + k - de] / nl2;
            v = (v + u) / 2.0f;
        }
        o[k] = v;
        acc += v * v;
    }
    if (!e2) return;   // a single view is already the concatenation of two unit vectors: the reference does not renormalise it
    const float nrm = fmaxf(sqrtf(block_sum(acc, sh)), 1e-12f);
    for (int k = tid; k < d; k += 256) o[k] = o[k] / nrm;
}

// gather rows idx[i] of x [.][d] into cur [ni][dp] (zero padded) and accumulate the column mean
__global__ void gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx, int ni, int d, int dp,
                                   float* __restrict__ cur) {
    const long long total = (long long)ni * dp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / dp), c = (int)(i - (long long)r * dp);
        cur[i] = c < d ? x[(long long)idx[r] * d + c] : 0.f;
    }
}
// column means of cur [ni][dp]: one thread per column, rows in order (the reference's mean(dim=0) sums rows sequentially)
__global__ void col_mean_kernel(const float* __restrict__ cur, int ni, int dp, float* __restrict__ mean) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= dp) return;
    float a = 0.f;
    for (int r = 0; r < ni; ++r) a += cur[(long long)r * dp + c];
    mean[c] = a / (float)ni;
}
__global__ void center_kernel(const float* __restrict__ cur, const float* __restrict__ mean, long long total, int dp,
                              float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        out[i] = cur[i] - mean[i % dp];
}
// cur [ni][dp] -> curT [dp][nip] (zero padded columns)
__global__ void transpose_kernel(const float* __restrict__ cur, int ni, int dp, int nip, float* __restrict__ curT) {
    const long long total = (long long)dp * nip;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i / nip), r = (int)(i - (long long)c * nip);
        curT[i] = r < ni ? cur[(long long)r * dp + c] : 0.f;
    }
}
// A = G + ridge * I on the real d x d block, ridge * I on the padding; row_abs[r] = sum |A[r][:]|
__global__ void ridge_rowsum_kernel(float* __restrict__ a, int d, int dp, float ridge, float* __restrict__ row_abs) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < dp; c += 256) {
        float v = a[(long long)r * dp + c];
        if (r >= d || c >= d) v = 0.f;
        if (r == c) v = r < d ? v + ridge : ridge;   // padding: any positive value inside the spectrum (never read back)
        a[(long long)r * dp + c] = v;
        s += fabsf(v);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) row_abs[r] = s;
}
__global__ void max_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ float sh[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, v[i]);
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__global__ void scaled_identity_kernel(const float* __restrict__ alpha, int dp, float* __restrict__ x) {
    const long long total = (long long)dp * dp;
    const float inv = 1.0f / alpha[0];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        x[i] = (i / dp == i % dp) ? inv : 0.f;
}
// tt <- (2I - t)^T, and res += ||I - t||_F^2 of t = A X (the residual of the current iterate).  The transposed copy makes
// the second product of the step, computed as X . tt^T by the A.B^T GEMM, exactly X (2I - A X).
__global__ void two_i_minus_t_kernel(const float* __restrict__ t, int dp, float* __restrict__ tt, float* __restrict__ res) {
    const long long total = (long long)dp * dp;
    float acc = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / dp), c = (int)(i - (long long)r * dp);
        const float eye = r == c ? 1.0f : 0.0f;
        const float v = t[i];
        acc += (eye - v) * (eye - v);
        tt[(long long)c * dp + r] = 2.0f * eye - v;
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(res, acc);
}
// x <- (x + x^T) / 2: keeps every iterate exactly symmetric, so that A . x^T (what the GEMM computes) is A x
__global__ void symmetrize_kernel(float* __restrict__ x, int dp) {
    const long long total = (long long)dp * dp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / dp), c = (int)(i - (long long)r * dp);
        if (c <= r) continue;
        const float m = 0.5f * (x[i] + x[(long long)c * dp + r]);
        x[i] = m;
        x[(long long)c * dp + r] = m;
    }
}
// out rows (normalised with torch.norm semantics: plain division) scattered back to x[idx[r]][:d]
__global__ __launch_bounds__(256) void normalize_scatter_kernel(const float* __restrict__ y, const int32_t* __restrict__ idx,
                                                                int d, int dp, float* __restrict__ x) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    float a = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) a += y[(long long)r * dp + c] * y[(long long)r * dp + c];
    const float nrm = sqrtf(block_sum(a, sh));
    float* o = x + (long long)idx[r] * d;
    for (int c = threadIdx.x; c < d; c += 256) o[c] = y[(long long)r * dp + c] / nrm;
}

inline int grid_for(long long work) {
    long long g = (work + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

int gemm_nt_dev(reid_ctx* ctx, const float* a, int m, const float* b, int n, int k, float* c) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = a; p.lda = k;
    p.B = b; p.ldb = k;
    p.M = m; p.N = n; p.K = k;
    p.C = c; p.ldc = n;
    return launch_gemm_f32(ctx, A_DENSE, E_BIAS, p, REID_K_DIST_GEMM, 2.0 * m * n * k, 4.0 * ((double)m * k + (double)n * k + (double)m * n));
}

}  // namespace

// x: [n][3][256][128] fp32 already normalised by the caller's transform (the evaluation script uses ImageNet mean/std,
// data_transforms.py:56-130); out: [n][512 + num_class].  flip_tta != 0 averages the descriptor of the mirrored image.
extern "C" int reid_descriptor_f32_nchw_dev(reid_ctx* ctx, const float* d_x, int n, int flip_tta, float* d_out) {
    ARG_CHECK(ctx && d_x && d_out && n >= 0);
    CTX_ENTER(ctx);
    if (n == 0) return REID_OK;
    int de = 0, nc = 0;
    REID_TRY(reid_seres18_dims(ctx, &de, &nc));
    if (nc <= 0) {
        reid_set_error("reid_descriptor_*: the loaded weights have no classifier (cls.w), the descriptor needs the logits");
        return REID_ERR_STATE;
    }
    const size_t img = (size_t)3 * 256 * 128;
    float *e1, *l1, *e2 = nullptr, *l2 = nullptr, *xf = nullptr;
    REID_TRY(ctx_ws(ctx, "pp.e1", (size_t)n * de * 4, (void**)&e1));
    REID_TRY(ctx_ws(ctx, "pp.l1", (size_t)n * nc * 4 + 16, (void**)&l1));
    REID_TRY(reid_embed_f32_nchw_dev(ctx, d_x, n, e1, l1));
    if (flip_tta) {
        REID_TRY(ctx_ws(ctx, "pp.e2", (size_t)n * de * 4, (void**)&e2));
        REID_TRY(ctx_ws(ctx, "pp.l2", (size_t)n * nc * 4 + 16, (void**)&l2));
        const int chunk = ctx->chunk;
        REID_TRY(ctx_ws(ctx, "pp.flip", (size_t)chunk * img * 4, (void**)&xf));
        for (int i = 0; i < n; i += chunk) {
            const int m = n - i < chunk ? n - i : chunk;
            const long long rows = (long long)m * 3 * 256;
            hipLaunchKernelGGL(flip_w_nchw_kernel, dim3(grid_for(rows * 128)), dim3(256), 0, ctx->stream, d_x + (size_t)i * img, rows,
                               128, xf);
            LAUNCH_CHECK();
            REID_TRY(reid_embed_f32_nchw_dev(ctx, xf, m, e2 + (size_t)i * de, l2 + (size_t)i * nc));
        }
    }
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n * (de + nc) * 4.0 * (flip_tta ? 3 : 2));
    hipLaunchKernelGGL(descriptor_kernel, dim3(n), dim3(256), 0, ctx->stream, e1, l1, e2, l2, de, nc, d_out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

extern "C" int reid_descriptor_f32_nchw(reid_ctx* ctx, const float* x, int n, int flip_tta, float* out) {
    ARG_CHECK(ctx && x && out && n >= 0);
    CTX_ENTER(ctx);
    if (n == 0) return REID_OK;
    int de = 0, nc = 0;
    REID_TRY(reid_seres18_dims(ctx, &de, &nc));
    const size_t img = (size_t)3 * 256 * 128;
    float *dx, *dout;
    REID_TRY(ctx_ws(ctx, "pp.x", (size_t)n * img * 4, (void**)&dx));
    REID_TRY(ctx_ws(ctx, "pp.out", (size_t)n * (de + (nc > 0 ? nc : 0)) * 4 + 16, (void**)&dout));
    HIP_TRY(hipMemcpyAsync(dx, x, (size_t)n * img * 4, hipMemcpyHostToDevice, ctx->stream));
    REID_TRY(reid_descriptor_f32_nchw_dev(ctx, dx, n, flip_tta, dout));
    HIP_TRY(hipMemcpyAsync(out, dout, (size_t)n * (de + nc) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ctx_fault_status(ctx);
}

// d_x [n][d] is updated in place; cams is a HOST array of camera ids >= 0 (ids without rows are skipped: the reference's
// torch.inverse of an all-zero matrix would raise).  iters <= 0 selects the default number of Newton-Schulz steps.
extern "C" int reid_cam_debias_dev(reid_ctx* ctx, float* d_x, const int32_t* cams, int n, int d, float la, int iters) {
    ARG_CHECK(ctx && d_x && cams && n >= 1 && d >= 1 && la > 0.f);
    CTX_GUARD(ctx);
    if (iters <= 0) iters = 40;   // upper bound; the residual rule below stops after ~8-14 steps
    int ncam = 0;
    for (int i = 0; i < n; ++i) {
        ARG_CHECK(cams[i] >= 0);
        if (cams[i] + 1 > ncam) ncam = cams[i] + 1;
    }
    const int dp = (d + 3) / 4 * 4;
    std::vector<std::vector<int32_t>> rows(ncam);
    for (int i = 0; i < n; ++i) rows[cams[i]].push_back(i);
    size_t max_ni = 0;
    for (auto& r : rows) max_ni = r.size() > max_ni ? r.size() : max_ni;
    const int max_nip = (int)((max_ni + 3) / 4 * 4);
    int32_t* d_idx;
    float *cur, *cen, *curT, *mean, *A, *X, *T, *T2, *X2, *rowabs, *alpha;
    REID_TRY(ctx_ws(ctx, "cd.idx", (size_t)n * 4, (void**)&d_idx));
    REID_TRY(ctx_ws(ctx, "cd.cur", max_ni * dp * 4, (void**)&cur));
    REID_TRY(ctx_ws(ctx, "cd.cen", max_ni * dp * 4, (void**)&cen));
    REID_TRY(ctx_ws(ctx, "cd.curT", (size_t)dp * max_nip * 4, (void**)&curT));
    REID_TRY(ctx_ws(ctx, "cd.mean", (size_t)dp * 4, (void**)&mean));
    REID_TRY(ctx_ws(ctx, "cd.A", (size_t)dp * dp * 4, (void**)&A));
    REID_TRY(ctx_ws(ctx, "cd.X", (size_t)dp * dp * 4, (void**)&X));
    REID_TRY(ctx_ws(ctx, "cd.T", (size_t)dp * dp * 4, (void**)&T));
    REID_TRY(ctx_ws(ctx, "cd.T2", (size_t)dp * dp * 4, (void**)&T2));
    REID_TRY(ctx_ws(ctx, "cd.X2", (size_t)dp * dp * 4, (void**)&X2));
    REID_TRY(ctx_ws(ctx, "cd.rowabs", (size_t)dp * 4, (void**)&rowabs));
    REID_TRY(ctx_ws(ctx, "cd.alpha", 16, (void**)&alpha));
    hipStream_t st = ctx->stream;
    size_t off = 0;
    std::vector<int32_t> flat;
    flat.reserve(n);
    for (auto& r : rows) flat.insert(flat.end(), r.begin(), r.end());
    HIP_TRY(hipMemcpyAsync(d_idx, flat.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int c = 0; c < ncam; ++c) {
        const int ni = (int)rows[c].size();
        if (ni == 0) continue;
        const int nip = (ni + 3) / 4 * 4;
        const int32_t* idx = d_idx + off;
        off += ni;
        const long long tot = (long long)ni * dp;
        hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(tot)), dim3(256), 0, st, d_x, idx, ni, d, dp, cur);
        hipLaunchKernelGGL(col_mean_kernel, dim3((dp + 255) / 256), dim3(256), 0, st, cur, ni, dp, mean);
        hipLaunchKernelGGL(center_kernel, dim3(grid_for(tot)), dim3(256), 0, st, cur, mean, tot, dp, cen);
        hipLaunchKernelGGL(transpose_kernel, dim3(grid_for((long long)dp * nip)), dim3(256), 0, st, cur, ni, dp, nip, curT);
        LAUNCH_CHECK();
        REID_TRY(gemm_nt_dev(ctx, curT, dp, curT, dp, nip, A));                      // G = X_c^T X_c (uncentred, as the reference)
        hipLaunchKernelGGL(ridge_rowsum_kernel, dim3(dp), dim3(256), 0, st, A, d, dp, (float)ni * la, rowabs);
        hipLaunchKernelGGL(max_kernel, dim3(1), dim3(256), 0, st, rowabs, dp, alpha);
        hipLaunchKernelGGL(scaled_identity_kernel, dim3(grid_for((long long)dp * dp)), dim3(256), 0, st, alpha, dp, X);
        LAUNCH_CHECK();
        // Newton-Schulz with a stopping rule: the residual ||I - A X||_F squares every step until it reaches the fp32 noise
        // floor; iterating past that point lets rounding errors (which do not commute with A) grow, so stop as soon as
        // the residual no longer halves and keep the better of the last two iterates.
        float *xa = X, *xb = X2;
        float r_prev = INFINITY;
        for (int it = 0; it < iters; ++it) {
            REID_TRY(gemm_nt_dev(ctx, A, dp, xa, dp, dp, T));                        // T = A X   (X symmetric)
            HIP_TRY(hipMemsetAsync(alpha + 1, 0, 4, st));
            hipLaunchKernelGGL(two_i_minus_t_kernel, dim3(grid_for((long long)dp * dp)), dim3(256), 0, st, T, dp, T2, alpha + 1);
            LAUNCH_CHECK();
            float r2 = 0.f;
            HIP_TRY(hipMemcpyAsync(&r2, alpha + 1, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const float r = sqrtf(r2);
            if (it == 0 && !(r < INFINITY)) {   // NaN / inf in the input rows: there is no previous iterate to fall back to
                reid_set_error("reid_cam_debias: non-finite residual for camera %d (NaN or inf in its rows?)", c);
                return REID_ERR_ARG;
            }
            if (!(r < r_prev)) {            // got worse (or NaN): the previous iterate is the answer
                float* t = xa; xa = xb; xb = t;
                break;
            }
            // quadratic phase (r_prev < 1e-2 would square to < 1e-4) but no longer halving: noise floor, keep the current iterate
            if ((r_prev < 1e-2f && r > 0.5f * r_prev) || r < 1e-6f) break;
            r_prev = r;
            REID_TRY(gemm_nt_dev(ctx, xa, dp, T2, dp, dp, xb));                      // X' = X (2I - A X)
            hipLaunchKernelGGL(symmetrize_kernel, dim3(grid_for((long long)dp * dp)), dim3(256), 0, st, xb, dp);
            LAUNCH_CHECK();
            float* t = xa; xa = xb; xb = t;
        }
        REID_TRY(gemm_nt_dev(ctx, cen, ni, xa, dp, dp, cur));                        // (X_c - mean) P^T, P symmetric
        hipLaunchKernelGGL(normalize_scatter_kernel, dim3(ni), dim3(256), 0, st, cur, idx, d, dp, d_x);
        LAUNCH_CHECK();
    }
    return REID_OK;
}

extern "C" int reid_cam_debias(reid_ctx* ctx, float* x, const int32_t* cams, int n, int d, float la, int iters) {
    ARG_CHECK(ctx && x && cams && n >= 1 && d >= 1);
    CTX_GUARD(ctx);
    float* dx;
    REID_TRY(ctx_ws(ctx, "cd.x", (size_t)n * d * 4, (void**)&dx));
    HIP_TRY(hipMemcpyAsync(dx, x, (size_t)n * d * 4, hipMemcpyHostToDevice, ctx->stream));
    REID_TRY(reid_cam_debias_dev(ctx, dx, cams, n, d, la, iters));
    HIP_TRY(hipMemcpyAsync(x, dx, (size_t)n * d * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}

// ---- smooth_tracklets, reid/inference_utils.py:18-27: every valid row of a tracklet (rows with the same sequence id) becomes
// keep * row + (1 - keep) * mean(valid rows of the tracklet), keep = 0.1 in the reference.  Rows are grouped on the host (CSR
// of row indices per tracklet); one block column-slice per tracklet: a thread owns a column, sums it over the tracklet's rows
// in row order (coalesced across the 256 columns of the slice), then rewrites the rows.
__global__ __launch_bounds__(256) void smooth_tracklets_kernel(float* __restrict__ x, const int32_t* __restrict__ rows,
                                                               const int32_t* __restrict__ start, int d, float keep) {
    const int g = blockIdx.x, col = blockIdx.y * 256 + threadIdx.x;
    if (col >= d) return;
    const int b = start[g], e = start[g + 1];
    float acc = 0.f;
    for (int i = b; i < e; ++i) acc += x[(long long)rows[i] * d + col];
    const float avg = acc / (float)(e - b);
    for (int i = b; i < e; ++i) {
        float* p = x + (long long)rows[i] * d + col;
        *p = *p * keep + avg * (1.0f - keep);
    }
}

extern "C" int reid_smooth_tracklets_dev(reid_ctx* ctx, float* d_x, const int32_t* seqs, const uint8_t* valid, int n, int d,
                                         float keep) {
    ARG_CHECK(ctx && d_x && seqs && n >= 0 && d >= 1);
    CTX_GUARD(ctx);
    if (n == 0) return REID_OK;
    std::map<int32_t, std::vector<int32_t>> groups;   // ordered like np.unique(seqs)
    for (int i = 0; i < n; ++i)
        if (!valid || valid[i]) groups[seqs[i]].push_back(i);
    if (groups.empty()) return REID_OK;
    std::vector<int32_t> rows, start(1, 0);
    for (auto& kv : groups) {
        rows.insert(rows.end(), kv.second.begin(), kv.second.end());
        start.push_back((int32_t)rows.size());
    }
    int32_t *d_rows, *d_start;
    REID_TRY(ctx_ws(ctx, "st.rows", rows.size() * 4, (void**)&d_rows));
    REID_TRY(ctx_ws(ctx, "st.start", start.size() * 4, (void**)&d_start));
    HIP_TRY(hipMemcpyAsync(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_start, start.data(), start.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(smooth_tracklets_kernel, dim3((unsigned)groups.size(), (d + 255) / 256), dim3(256), 0, ctx->stream, d_x, d_rows,
                       d_start, d, keep);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // rows / start are locals
    return REID_OK;
}

extern "C" int reid_smooth_tracklets(reid_ctx* ctx, float* x, const int32_t* seqs, const uint8_t* valid, int n, int d, float keep) {
    ARG_CHECK(ctx && x && seqs && n >= 0 && d >= 1);
    CTX_GUARD(ctx);
    if (n == 0) return REID_OK;
    float* dx;
    REID_TRY(ctx_ws(ctx, "cd.x", (size_t)n * d * 4, (void**)&dx));
    HIP_TRY(hipMemcpyAsync(dx, x, (size_t)n * d * 4, hipMemcpyHostToDevice, ctx->stream));
    REID_TRY(reid_smooth_tracklets_dev(ctx, dx, seqs, valid, n, d, keep));
    HIP_TRY(hipMemcpyAsync(x, dx, (size_t)n * d * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}
