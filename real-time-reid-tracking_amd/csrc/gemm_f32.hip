// C[M][N] = A'[M][K] . B[N][K]^T on v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain).
//
// One kernel template serves every dense contraction of the hot path:
//   * the 3x3 / 1x1 convolutions of the SE blocks  (A' = im2col gather of NHWC activations, fused
//     per-(image,channel) affine+ReLU on the way in = InstanceNorm/BatchNorm of the previous conv,
//     SERes18_IBN.py:88-93; fused BN / residual / ReLU / per-channel partial sums on the way out)
//   * the 7x7 stem                                  (A' gathered from fp32 or straight from uint8 crops)
//   * the N x M distance matrix                     (reid/losses/utils.py:12-35 epilogue)
//   * Linear layers                                 (classifier, SERes18_IBN.py:271)
//
// Tiling: 256 threads = 4 waves in a 2x2 grid, block tile 128 x BN (BN = 64 or 128), BK = 32.
// Each wave owns 64 x BN/2 outputs = 2 x (BN/64) MFMA tiles of 32x32, accumulators stay in registers.
// LDS rows are padded to 36 floats: the ds_read_b128 of 16 different rows then hits 16 distinct 4-bank
// groups (36*i mod 64 is a distinct multiple of 4 for the lane groups of ds_read_b128) -> conflict free.
// Operands: lane (i = lane&31, h = lane>>5) reads 4 consecutive k (4h..4h+3) of row i with ONE b128 read and
// feeds them to 4 MFMAs; both operands use the same k permutation, so each MFMA sums k in {e, 4+e}.
// Global -> register -> LDS staging is split (loads for tile t+1 are issued before the MFMAs of tile t and
// written to LDS after them), so HBM/L2 latency hides under 64 MFMAs x 64 cycles per wave.
// blockIdx is remapped so that the blocks that share an XCD (b % 8) cover consecutive (m-tile, n-tile)
// pairs: the N tiles of one M tile and neighbouring M tiles (conv halo) then share that XCD's L2.
#include "reid_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDP = 36;  // padded LDS row pitch in floats

template <int AMODE>
struct ARows {
    int img[4], iy0[4], ix0[4];
    bool ok[4];
    int m[4];
};

template <int AMODE>
__device__ __forceinline__ void load_a_tile(const GemmParams& p, const ARows<AMODE>& rw, int kt, int c4, f32x4 (&ra)[4]) {
    const int k0 = kt * BK;
    if constexpr (AMODE == A_DENSE) {
        const int k = k0 + c4 * 4;
        const float* A = (const float*)p.A;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (rw.ok[j] && k < p.K) v = *(const f32x4*)(A + (long long)rw.m[j] * p.lda + k);
            ra[j] = v;
        }
    } else if constexpr (AMODE == A_IM2COL) {
        const int tap = k0 / p.Cin;
        const int c = k0 - tap * p.Cin + c4 * 4;
        const int r = tap / p.S;
        const int s = tap - r * p.S;
        const float* A = (const float*)p.A;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = rw.iy0[j] + r, ix = rw.ix0[j] + s;
            const bool ok = rw.ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                v = *(const f32x4*)(A + (((long long)rw.img[j] * p.H + iy) * p.W + ix) * p.Cin + c);
                if (p.a_scale) {
                    const f32x4 sc = *(const f32x4*)(p.a_scale + (long long)rw.img[j] * p.Cin + c);
                    const f32x4 sh = *(const f32x4*)(p.a_shift + (long long)rw.img[j] * p.Cin + c);
                    v = v * sc + sh;
                    if (p.a_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                }
            }
            ra[j] = v;
        }
    } else {  // stem: k = r*24 + (s*3 + c); the 21 taps of one kernel row are contiguous in NHWC memory
        const int k = k0 + c4 * 4;
        const int r = k / 24;
        const int j0 = k - r * 24;
        const int w3 = p.W * 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = rw.iy0[j] + r;
            const bool rowok = rw.ok[j] && r < 7 && (unsigned)iy < (unsigned)p.H;
            const long long base = ((long long)rw.img[j] * p.H + iy) * w3;
            const int e0 = rw.ix0[j] * 3 + j0;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = e0 + e;
                const bool ok = rowok && (j0 + e) < 21 && idx >= 0 && idx < w3;
                float x = 0.f;
                if (ok) {
                    if constexpr (AMODE == A_STEM_U8) {
                        // feature_extractor.py:41-46: im.astype(float32)/255 -> Normalize(0.5, 0.5)
                        x = ((float)((const uint8_t*)p.A)[base + idx] / 255.0f - 0.5f) / 0.5f;
                    } else {
                        x = ((const float*)p.A)[base + idx];
                    }
                }
                v[e] = x;
            }
            ra[j] = v;
        }
    }
}

template <int AMODE, int EPI, int BN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
    constexpr int TM = 2;
    constexpr int TN = BN / 64;
    constexpr int BCH = BN / 32;  // 16-byte chunks of the B tile per thread
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDP];
    float* As = lds;
    float* Bs = lds + BM * LDP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware, bijective block remap (blocks b and b+8 share an XCD)
    const int nnt = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int mtile, ntile;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        mtile = L / nnt;
        ntile = L - mtile * nnt;
    }
    const int m_blk = mtile * BM, n_blk = ntile * BN;

    const int c4 = tid & 7, lrow = tid >> 3;
    ARows<AMODE> rw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m_blk + j * 32 + lrow;
        rw.m[j] = m;
        rw.ok[j] = m < p.M;
        if constexpr (AMODE != A_DENSE) {
            const int hw = p.Ho * p.Wo;
            const int img = m / hw, rem = m - img * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            rw.img[j] = img;
            rw.iy0[j] = oy * p.stride - p.pad_y;
            rw.ix0[j] = ox * p.stride - p.pad_x;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    f32x4 ra[4], rb[BCH];
    auto load_b = [&](int kt) {
        const int k = kt * BK + c4 * 4;
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            const int n = n_blk + j * 32 + lrow;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < p.N && k < p.K) v = *(const f32x4*)(p.B + (long long)n * p.ldb + k);
            rb[j] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)&As[(j * 32 + lrow) * LDP + c4 * 4] = ra[j];
#pragma unroll
        for (int j = 0; j < BCH; ++j) *(f32x4*)&Bs[(j * 32 + lrow) * LDP + c4 * 4] = rb[j];
    };

    const int nk = (p.K + BK - 1) / BK;
    load_a_tile<AMODE>(p, rw, 0, c4, ra);
    load_b(0);
    store_tiles();
    __syncthreads();

    const float* a_rd = As + (wm * 64 + li) * LDP + lh * 4;
    const float* b_rd = Bs + (wn * (BN / 2) + li) * LDP + lh * 4;

    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            load_a_tile<AMODE>(p, rw, kt + 1, c4, ra);
            load_b(kt + 1);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *(const f32x4*)(a_rd + a * 32 * LDP + kk * 8);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = *(const f32x4*)(b_rd + b * 32 * LDP + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        if (more) store_tiles();
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* stat_lds = lds;  // [2 (wm)][BN][2], reused after the last barrier
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int lcol = wn * (BN / 2) + b * 32 + li;
        const int col = n_blk + lcol;
        const bool colok = col < p.N;
        float cs = 1.f, sh = 0.f, cq = 0.f;
        if constexpr (EPI == E_CONV) {
            if (colok && p.col_scale) { cs = p.col_scale[col]; sh = p.col_shift[col]; }
        } else if constexpr (EPI == E_BIAS) {
            if (colok && p.col_shift) sh = p.col_shift[col];
        } else {
            if (colok && p.col_sq) cq = p.col_sq[col];
            if (p.metric == REID_METRIC_COS_HALF || p.metric == REID_METRIC_COS) cq = sqrtf(cq);
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m_blk + wm * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const bool ok = colok && row < p.M;
                long long orow = row;
                if constexpr (EPI == E_BIAS) {
                    if (p.scat_h > 0) {   // ConvTranspose2d(4,2,1) output parity: (img, j, i) -> (img, 2j+py, 2i+px)
                        const int hw = p.scat_h * p.scat_w;
                        const int img = row / hw, rem = row - img * hw;
                        const int j = rem / p.scat_w, i = rem - j * p.scat_w;
                        orow = ((long long)img * 2 * p.scat_h + 2 * j + p.scat_py) * (2 * p.scat_w) + 2 * i + p.scat_px;
                    }
                }
                const long long idx = orow * p.ldc + col;
                float v = acc[a][b][e];
                if constexpr (EPI == E_CONV) {
                    if (p.col_scale) v = v * cs + sh;
                    if (p.residual && ok) v += p.residual[idx];
                    if (p.relu && col >= p.relu_from) v = fmaxf(v, 0.f);
                    if (ok) { s1 += v; s2 += v * v; }
                } else if constexpr (EPI == E_BIAS) {
                    v += sh;
                    if (p.act == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));   // nn.GELU() (erf form)
                } else {
                    float rs = 0.f;
                    if (ok && p.row_sq) rs = p.row_sq[row];
                    switch (p.metric) {
                        case REID_METRIC_L2: v = sqrtf(fmaxf(l2sqr_of(v, rs, cq), 1e-12f)); break;
                        case REID_METRIC_L2SQR: v = l2sqr_of(v, rs, cq); break;
                        case REID_METRIC_COS_HALF: v = (1.0f - v / (sqrtf(rs) * cq)) / 2.0f; break;
                        case REID_METRIC_COS: v = 1.0f - v / (sqrtf(rs) * cq); break;
                        default: break;
                    }
                }
                if constexpr (EPI == E_BIAS) {
                    if (p.residual && ok) v += p.residual[idx];
                }
                if (ok) p.C[idx] = v;
            }
        }
        if constexpr (EPI == E_CONV) {
            if (p.stats) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (lh == 0) {
                    stat_lds[(wm * BN + lcol) * 2 + 0] = s1;
                    stat_lds[(wm * BN + lcol) * 2 + 1] = s2;
                }
            }
        }
    }
    if constexpr (EPI == E_CONV) {
        if (p.stats) {
            __syncthreads();
            if (tid < BN && n_blk + tid < p.N) {
                const float t1 = stat_lds[tid * 2 + 0] + stat_lds[(BN + tid) * 2 + 0];
                const float t2 = stat_lds[tid * 2 + 1] + stat_lds[(BN + tid) * 2 + 1];
                float* o = p.stats + ((long long)mtile * p.N + n_blk + tid) * 2;
                o[0] = t1;
                o[1] = t2;
            }
        }
    }
}

template <int AMODE, int EPI>
int launch_bn(reid_ctx* ctx, const GemmParams& p) {
    const int nmt = (p.M + BM - 1) / BM;
    if (p.N <= 64) {
        const int nnt = (p.N + 63) / 64;
        hipLaunchKernelGGL((gemm_f32_kernel<AMODE, EPI, 64>), dim3(nmt * nnt), dim3(256), 0, ctx->stream, p);
    } else {
        const int nnt = (p.N + 127) / 128;
        hipLaunchKernelGGL((gemm_f32_kernel<AMODE, EPI, 128>), dim3(nmt * nnt), dim3(256), 0, ctx->stream, p);
    }
    LAUNCH_CHECK();
    return REID_OK;
}

}  // namespace

int launch_gemm_f32(reid_ctx* ctx, int amode, int epi, const GemmParams& p, int kind, double flops, double bytes) {
    ARG_CHECK(p.M > 0 && p.N > 0 && p.K > 0);
    ARG_CHECK(p.K % 4 == 0 && p.ldb % 4 == 0);
    if (amode == A_DENSE) ARG_CHECK(p.lda % 4 == 0);
    if (amode == A_IM2COL) ARG_CHECK(p.Cin % 32 == 0 && p.K == p.R * p.S * p.Cin);
    if (amode == A_STEM_F32 || amode == A_STEM_U8) ARG_CHECK(p.K == 192 && p.Cin == 3);
    if (epi == E_CONV && p.stats) ARG_CHECK(p.M % BM == 0);
    // dense contractions (distance matrix, Linear layers): the LDS-DMA kernel (REID_F32_CONV=0 / 2 keep this file's kernel for A/B)
    if (ctx->f32_conv == 1 && gemm_f32_dma_supported(amode, epi, p)) return launch_gemm_f32_dma(ctx, epi, p, kind, flops, bytes);
    prof_begin(ctx, kind, flops, bytes);
    int st = REID_ERR_ARG;
    if (amode == A_DENSE && epi == E_DIST) st = launch_bn<A_DENSE, E_DIST>(ctx, p);
    else if (amode == A_DENSE && epi == E_BIAS) st = launch_bn<A_DENSE, E_BIAS>(ctx, p);
    else if (amode == A_IM2COL && epi == E_CONV) st = launch_bn<A_IM2COL, E_CONV>(ctx, p);
    else if (amode == A_IM2COL && epi == E_BIAS) st = launch_bn<A_IM2COL, E_BIAS>(ctx, p);
    else if (amode == A_STEM_F32 && epi == E_CONV) st = launch_bn<A_STEM_F32, E_CONV>(ctx, p);
    else if (amode == A_STEM_U8 && epi == E_CONV) st = launch_bn<A_STEM_U8, E_CONV>(ctx, p);
    else reid_set_error("launch_gemm_f32: unsupported (amode=%d, epi=%d)", amode, epi);
    prof_end(ctx);
    return st;
}
