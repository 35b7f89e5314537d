// C[M][N] (f16) = A'[M][K] (f16) . B[N][K]^T (f16), fp32 accumulate, on v_mfma_f32_32x32x16_f16.
// The fast path of the convolution stack (reid_ctx_set_precision(ctx, 1)): activations and weights are stored as
// fp16, every sum is fp32, BN / residual / ReLU / statistics run in fp32 in the epilogue.
//
// Staging is LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, the im2col gather is the per-lane SOURCE
// address, zero padding is a lane pointing at a zero page.  The LDS image is lane-linear ([row][8 chunks of 16 B],
// 128-B rows), so the bank-conflict fix is an XOR swizzle applied to the source chunk and to the read address:
// chunk c of row r sits at position c ^ ((r >> 1) & 7).  For the lane groups of ds_read_b128 the 16 rows of a group
// then cover 16 distinct 16-B slots of the 256-B bank row (conflict-free).
// Two LDS stages: the DMA of K-tile t+1 is issued before the MFMAs of tile t; one barrier per K-tile.
// Tiling as the f32 kernel: 4 waves (2x2), block 128 x BN, wave 64 x BN/2, BK = 64.
// Requirements (all met by the conv stack): M % 128 == 0, N % BN == 0, K % 64 == 0.
#include "reid_internal.h"

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(uintptr_t)(p))

template <int AMODE, int BN>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(const Gemm16Params p) {
    constexpr int TM = 2, TN = BN / 64;
    constexpr int A_BYTES = 128 * 128;  // 128 rows x 64 f16
    constexpr int B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int BJ = BN / 32;         // B wave-instructions per wave per K-tile
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int nnt = p.N / BN;
    const int nwg = gridDim.x;
    int mtile, ntile;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        mtile = L / nnt;
        ntile = L - mtile * nnt;
    }
    const int m_blk = mtile * 128, n_blk = ntile * BN;

    // ---- per-lane source descriptors: wave-instruction j of this wave fills tile rows (wave*4+j)*8 .. +8
    int a_chunk[4];
    long long a_base[4];  // dense: element offset of the row
    int a_img[4], a_iy0[4], a_ix0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + (lane >> 3);
        a_chunk[j] = (lane & 7) ^ ((row >> 1) & 7);
        const int m = m_blk + row;
        if constexpr (AMODE == A16_DENSE) {
            a_base[j] = (long long)m * p.lda;
        } else {
            const int hw = p.Ho * p.Wo;
            const int img = m / hw, rem = m - img * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_iy0[j] = oy * p.stride - p.pad;
            a_ix0[j] = ox * p.stride - p.pad;
            a_img[j] = img;
        }
    }
    int b_chunk[BJ];
    long long b_base[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = wave * (BN / 4) + j * 8 + (lane >> 3);
        b_chunk[j] = (lane & 7) ^ ((row >> 1) & 7);
        b_base[j] = (long long)(n_blk + row) * p.ldb;
    }

    auto stage = [&](int kt, int buf) {
        char* As = lds + buf * STAGE;
        char* Bs = As + A_BYTES;
        const int k0 = kt * 64;
        if constexpr (AMODE == A16_DENSE) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds(GPTR(p.A + a_base[j] + k0 + a_chunk[j] * 8),
                                                 LPTR(As + (wave * 32 + j * 8) * 128), 16, 0, 0);
        } else if constexpr (AMODE == A16_IM2COL) {
            const int tap = k0 / p.Cin;
            const int c0 = k0 - tap * p.Cin;
            const int r = tap / p.S, s = tap - r * p.S;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = a_iy0[j] + r, ix = a_ix0[j] + s;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const f16* src = ok ? p.A + (((long long)a_img[j] * p.H + iy) * p.W + ix) * p.Cin + c0 + a_chunk[j] * 8
                                    : p.zero_page;
                __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(As + (wave * 32 + j * 8) * 128), 16, 0, 0);
            }
        } else {  // A16_STEM: zero-padded NHWC4 input, k = r*32 + s*4 + c, one K-tile = kernel rows 2kt, 2kt+1
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 2 * kt + (a_chunk[j] >> 2);
                const int sp = a_chunk[j] & 3;
                // a_iy0 = 2*oy - 3, padded row index = a_iy0 + 3 + r; same for columns (pairs of pixels)
                const long long pix = ((long long)a_img[j] * p.Hp + (a_iy0[j] + 3 + r)) * p.Wp + (a_ix0[j] + 3 + 2 * sp);
                __builtin_amdgcn_global_load_lds(GPTR(p.A + pix * 4), LPTR(As + (wave * 32 + j * 8) * 128), 16, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            __builtin_amdgcn_global_load_lds(GPTR(p.B + b_base[j] + k0 + b_chunk[j] * 8),
                                             LPTR(Bs + (wave * (BN / 4) + j * 8) * 128), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int nk = p.K / 64;
    const int swz = (li >> 1) & 7;
    const int a_row_off = (wm * 64 + li) * 128;
    const int b_row_off = (wn * (BN / 2) + li) * 128;

    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* As = lds + buf * STAGE;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int pos = ((kk * 2 + lh) ^ swz) * 16;
            half8 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *(const half8*)(As + a_row_off + a * 32 * 128 + pos);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = *(const half8*)(Bs + b_row_off + b * 32 * 128 + pos);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();  // drains this tile's DMA (vmcnt(0)) and orders reads of `buf` before its next refill
    }

    // ------------------------------------------------------------------ epilogue (fp32 math, f16 stores)
    float* stat_lds = (float*)lds;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int lcol = wn * (BN / 2) + b * 32 + li;
        const int col = n_blk + lcol;
        float cs = 1.f, sh = 0.f;
        if (p.col_scale) { cs = p.col_scale[col]; sh = p.col_shift[col]; }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m_blk + wm * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const long long idx = (long long)row * p.ldc + col;
                float v = acc[a][b][e];
                if (p.col_scale) v = v * cs + sh;
                if (p.residual) v += (float)p.residual[idx];
                if (p.relu) v = fmaxf(v, 0.f);
                s1 += v;
                s2 += v * v;
                p.C[idx] = (f16)v;
            }
        }
        if (p.stats) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lh == 0) {
                stat_lds[(wm * BN + lcol) * 2 + 0] = s1;
                stat_lds[(wm * BN + lcol) * 2 + 1] = s2;
            }
        }
    }
    if (p.stats) {
        __syncthreads();
        if (tid < BN) {
            float* o = p.stats + ((long long)mtile * p.N + n_blk + tid) * 2;
            o[0] = stat_lds[tid * 2 + 0] + stat_lds[(BN + tid) * 2 + 0];
            o[1] = stat_lds[tid * 2 + 1] + stat_lds[(BN + tid) * 2 + 1];
        }
    }
}

template <int AMODE>
int launch_bn16(reid_ctx* ctx, const Gemm16Params& p) {
    const int nmt = p.M / 128;
    if (p.N % 128 == 0) {
        hipLaunchKernelGGL((gemm_f16_kernel<AMODE, 128>), dim3(nmt * (p.N / 128)), dim3(256), 0, ctx->stream, p);
    } else {
        hipLaunchKernelGGL((gemm_f16_kernel<AMODE, 64>), dim3(nmt * (p.N / 64)), dim3(256), 0, ctx->stream, p);
    }
    LAUNCH_CHECK();
    return REID_OK;
}

}  // namespace

int launch_gemm_f16(reid_ctx* ctx, int amode, const Gemm16Params& p, int kind, double flops, double bytes) {
    ARG_CHECK(p.M > 0 && p.M % 128 == 0 && p.N % 64 == 0 && p.K % 64 == 0 && p.ldb % 8 == 0);
    if (amode == A16_IM2COL) ARG_CHECK(p.Cin % 64 == 0 && p.K == p.R * p.S * p.Cin && p.zero_page);
    if (amode == A16_STEM) ARG_CHECK(p.K == 256 && p.Hp >= p.H + 6 && p.Wp >= p.W + 8);
    if (amode == A16_DENSE) ARG_CHECK(p.lda % 8 == 0);
    prof_begin(ctx, kind, flops, bytes);
    int st = REID_ERR_ARG;
    if (amode == A16_IM2COL) st = launch_bn16<A16_IM2COL>(ctx, p);
    else if (amode == A16_STEM) st = launch_bn16<A16_STEM>(ctx, p);
    else if (amode == A16_DENSE) st = launch_bn16<A16_DENSE>(ctx, p);
    else reid_set_error("launch_gemm_f16: unsupported amode %d", amode);
    prof_end(ctx);
    return st;
}
