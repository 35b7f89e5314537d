// Convolutions of the exact-fp32 path (SERes18_IBN.py:88-128, 250-276) as implicit GEMM on v_mfma_f32_32x32x2_f32.
//
//   C[M][Cout] = im2col(A)[M][R*S*Cin] . W[Cout][R*S*Cin]^T,   M = n*Ho*Wo, NHWC fp32 activations, k = (r*S+s)*Cin + c
//
// Same arithmetic as gemm_f32_kernel<A_IM2COL, E_CONV> (a k-ordered fmaf chain per output, bit for bit), rebuilt around
// what bounded that kernel (profiles/r01: 94 TF/s = 0.60 of the 157 TF fp32 MFMA peak):
//   * every global load of a K-tile is issued UNCONDITIONALLY (padding taps read a clamped in-image address and are
//     zeroed when the tile is written to LDS), so nothing waits inside a branch: the old loader ran four serialized
//     L2 round trips per K-tile in front of the MFMAs (load, branch, load scale/shift, s_waitcnt vmcnt(0), transform);
//   * the InstanceNorm/BatchNorm + ReLU of the previous conv (per-(image, channel) scale/shift) is applied when the
//     staged registers go to LDS, one K-tile after the loads were issued, i.e. behind 64 MFMAs x 64 cycles;
//   * two LDS buffers: the next tile is written between the two halves of this tile's MFMAs, one barrier per K-tile;
//   * a block is one (128-row, BN-column) tile and 128 rows never straddle an image (Ho*Wo % 128 == 0), so the
//     scale/shift vectors, the image base and all tap arithmetic are wave-uniform scalars;
//   * the epilogue has no per-element predicates: tiles are full (M % 128 == 0, Cout % BN == 0), residual values are
//     fetched 16 at a time before they are needed.
// Tiling: 256 threads = 4 waves (2 x 2), wave tile 64 x BN/2, BK = 32, LDS rows padded to 36 floats (conflict-free
// ds_read_b128 of 16 different rows), XCD-aware block order (the N tiles of an M tile and neighbouring M tiles share an L2).
#include "reid_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDP = 36;

template <int BN, bool AFF>
__global__ __launch_bounds__(256, 2) void conv_f32_kernel(const GemmParams p) {
    constexpr int WN = BN / 2;      // wave tile width
    constexpr int TM = 2;
    constexpr int TN = WN / 32;
    constexpr int BCH = BN / 32;    // 16-byte chunks of the weight tile per thread
    constexpr int TILE = (BM + BN) * LDP;
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int nnt = p.N / BN;
    int mtile, ntile;
    {
        const int nwg = gridDim.x;
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        mtile = L / nnt;
        ntile = L - mtile * nnt;
    }
    const int m_blk = mtile * BM, n_blk = ntile * BN;
    const int hw = p.Ho * p.Wo;
    const int img = m_blk / hw;                     // uniform: a tile lies inside one image
    const int rem_blk = m_blk - img * hw;

    const int c4 = tid & 7, lrow = tid >> 3;
    int iy0[4], ix0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rem = rem_blk + j * 32 + lrow;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        iy0[j] = oy * p.stride - p.pad_y;
        ix0[j] = ox * p.stride - p.pad_x;
    }
    const float* Aimg = (const float*)p.A + (long long)img * p.H * p.W * p.Cin + c4 * 4;
    const float* Bthr = p.B + (long long)(n_blk + lrow) * p.ldb + c4 * 4;
    const float* sc_ptr = AFF ? p.a_scale + (long long)img * p.Cin + c4 * 4 : nullptr;
    const float* sh_ptr = AFF ? p.a_shift + (long long)img * p.Cin + c4 * 4 : nullptr;
    const int cpt = p.Cin / BK;                     // K-tiles per tap
    const int nk = p.R * p.S * cpt;

    // loader state (uniform): the K-tile the next issue_loads() fetches
    int l_r = 0, l_s = 0, l_cc = 0, l_k = 0;
    f32x4 ra[4], rb[BCH], rsc, rsh;
    unsigned okm = 0;
    auto issue_loads = [&]() {
        const int c0 = l_cc * BK;
        okm = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = iy0[j] + l_r, ix = ix0[j] + l_s;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
            ra[j] = *(const f32x4*)(Aimg + (iyc * p.W + ixc) * p.Cin + c0);
            okm |= (ok ? 1u : 0u) << j;
        }
        if constexpr (AFF) {
            rsc = *(const f32x4*)(sc_ptr + c0);
            rsh = *(const f32x4*)(sh_ptr + c0);
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) rb[j] = *(const f32x4*)(Bthr + (long long)j * 32 * p.ldb + l_k);
        l_k += BK;
        if (++l_cc == cpt) {
            l_cc = 0;
            if (++l_s == p.S) { l_s = 0; ++l_r; }
        }
    };
    const float a_lo = p.a_relu ? 0.f : -3.402823466e38f;   // ReLU as a max against 0 or -FLT_MAX: no branch in the tile writer
    auto store_tile = [&](float* buf) {
        float* As = buf;
        float* Bs = buf + BM * LDP;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = ra[j];
            if constexpr (AFF) {
                v = v * rsc + rsh;
                v.x = fmaxf(v.x, a_lo); v.y = fmaxf(v.y, a_lo); v.z = fmaxf(v.z, a_lo); v.w = fmaxf(v.w, a_lo);
            }
            if (!((okm >> j) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *(f32x4*)&As[(j * 32 + lrow) * LDP + c4 * 4] = v;
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) *(f32x4*)&Bs[(j * 32 + lrow) * LDP + c4 * 4] = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int a_off = (wm * 64 + li) * LDP + lh * 4;
    const int b_off = BM * LDP + (wn * WN + li) * LDP + lh * 4;
    auto mfma_half = [&](const float* buf, int half) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int kk = half * 2 + k2;
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *(const f32x4*)(buf + a_off + a * 32 * LDP + kk * 8);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = *(const f32x4*)(buf + b_off + b * 32 * LDP + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
        }
    };

    issue_loads();
    store_tile(lds);
    if (nk > 1) issue_loads();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        float* cur = lds + (kt & 1) * TILE;
        float* nxt = lds + ((kt + 1) & 1) * TILE;
        mfma_half(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            store_tile(nxt);                  // tile kt+1: its loads were issued one K-tile ago
            if (kt + 2 < nk) issue_loads();   // tile kt+2: lands behind the next 64 MFMAs
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(cur, 1);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue (full tiles, no predicates)
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* stat_lds = lds;   // [2 (wm)][BN][2]; every wave is past the last barrier, nothing reads the tiles any more
    const bool has_cs = p.col_scale != nullptr;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int lcol = wn * WN + b * 32 + li;
        const int col = n_blk + lcol;
        const float cs = has_cs ? p.col_scale[col] : 1.f;
        const float sh = has_cs ? p.col_shift[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const long long base = (long long)(m_blk + wm * 64 + a * 32 + 4 * lh) * p.ldc + col;
            float res[16];
            if (p.residual) {
#pragma unroll
                for (int e = 0; e < 16; ++e) res[e] = p.residual[base + (long long)((e & 3) + 8 * (e >> 2)) * p.ldc];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[a][b][e];
                if (has_cs) v = v * cs + sh;
                if (p.residual) v += res[e];
                if (p.relu) v = fmaxf(v, 0.f);
                s1 += v;
                s2 += v * v;
                p.C[base + (long long)((e & 3) + 8 * (e >> 2)) * p.ldc] = v;
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
            const float t1 = stat_lds[tid * 2 + 0] + stat_lds[(BN + tid) * 2 + 0];
            const float t2 = stat_lds[tid * 2 + 1] + stat_lds[(BN + tid) * 2 + 1];
            float* o = p.stats + ((long long)mtile * p.N + n_blk + tid) * 2;
            o[0] = t1;
            o[1] = t2;
        }
    }
}

template <int BN>
void launch_bn(reid_ctx* ctx, const GemmParams& p) {
    const int grid = (p.M / BM) * (p.N / BN);
    if (p.a_scale) hipLaunchKernelGGL((conv_f32_kernel<BN, true>), dim3(grid), dim3(256), 0, ctx->stream, p);
    else hipLaunchKernelGGL((conv_f32_kernel<BN, false>), dim3(grid), dim3(256), 0, ctx->stream, p);
}

}  // namespace

bool conv_f32_supported(const GemmParams& p) {
    return p.Cin % BK == 0 && p.K == p.R * p.S * p.Cin && p.M % BM == 0 && (p.Ho * p.Wo) % BM == 0 && p.N % 64 == 0 &&
           p.ldb % 4 == 0 && p.ldc == p.N && (p.a_scale == nullptr) == (p.a_shift == nullptr) &&
           (p.col_scale == nullptr) == (p.col_shift == nullptr) && (long long)p.H * p.W * p.Cin < (1ll << 31);
}

int launch_conv_f32(reid_ctx* ctx, const GemmParams& p, int kind, double flops, double bytes) {
    ARG_CHECK(conv_f32_supported(p));
    prof_begin(ctx, kind, flops, bytes);
    if (p.N % 128 == 0) launch_bn<128>(ctx, p);
    else launch_bn<64>(ctx, p);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
