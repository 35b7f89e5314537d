// C[M][N] = A[M][K] . B[N][K]^T (dense, row-major, K contiguous) on v_mfma_f32_32x32x2_f32 with LDS-DMA staging - the
// dense sibling of conv_f32_dma_kernel (conv_f32.hip) for
//   * the N x M distance matrix (reid/losses/utils.py:12-35, reid/evaluate.py:58): distance epilogue,
//   * the Swin Linear layers in exact-fp32 mode (swin_transformer.py:23-39, 191-232): bias / erf-GELU / residual epilogue,
//   * the classifier (SERes18_IBN.py:271).
// Same reasoning as there: beside back-to-back fp32 MFMAs a SIMD issues one VALU instruction per ~19 cycles, so the loader
// must not cost VALU.  A K-tile is 8 x buffer_load_dwordx4 ... lds per wave with per-lane byte offsets fixed for the whole
// kernel and the K offset in the scalar operand; ragged edges need no predicates in the loop: the buffer descriptors end at
// row M of A / row N of B, so rows past the edge read as zeros.  128 x {64,128} tile, 4 waves, BK = 32, two LDS stages,
// XOR-swizzled 128-byte rows (conflict-free ds_read_b128), one s_waitcnt vmcnt(0) + one raw s_barrier per K-tile.
// Requirements: K % 32 == 0, 16-byte aligned rows (lda, ldb % 4 == 0); everything else falls back to gemm_f32_kernel.
#include "reid_internal.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128;
#define LPTR(p) ((__attribute__((address_space(3))) void*)(uintptr_t)(p))
#define RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

// One output tile's operand descriptors: they start at the tile's first row and end at the matrix' last row, so rows past
// M / N read as zeros (and a row past the edge keeps an offset beyond the descriptor for every K-tile).
template <int AJ, int BJ>
struct TileDesc {
    __amdgpu_buffer_rsrc_t a_rs, b_rs;
    int a_voff[AJ], b_voff[BJ];
    int m_blk, n_blk, rows_a;
};

// BK = 32: 128-byte LDS rows, eight 16-byte chunks, two blocks per CU (64 KB of LDS for the 128-wide tile).
// BK = 16 (the 128-wide distance tile): 64-byte rows, four chunks, 32 KB of LDS and <= 168 VGPRs -> THREE blocks per CU.  Why: s_memtime
// stamps of the Market-size matrix (3368 x 15913 x 512, profiles/r05_experiments.txt) - a block's epilogue (64 x [3 FMA, max, sqrt,
// 4-byte store] per lane, issued beside the other block's back-to-back MFMAs) lasts LONGER than its 16-tile K loop (83-103 k against 74 k
// cycles), so with two blocks per CU a SIMD has ONE wave feeding the matrix pipe most of the time; a third block keeps two.
template <int BN, int EPI, int BK = 32>
__global__ __launch_bounds__(256, BK == 16 ? 3 : 2) void gemm_f32_dma_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WN = BN / 2, TM = 2, TN = WN / 32;
    constexpr int ROWB = BK * 4;
    constexpr int CH = ROWB / 16;                    // 16-byte chunks per LDS row: 8 or 4
    constexpr int RPP = 1024 / ROWB;                 // rows per 1-KB DMA piece: 8 or 16
    constexpr int KK = BK / 8;                       // ds_read_b128 steps per K-tile (a step = chunks 2 kk + lh: four MFMAs per tile pair)
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int AJ = BM / RPP / 4, BJ = BN / RPP / 4;
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int nnt = (p.N + BN - 1) / BN;
    const int nk = p.K / BK;

    // XOR swizzle of the chunk position: the 16 rows a ds_read_b128 phase touches (16 lanes, one chunk each) must cover the 16
    // bank groups - rows are 128 B (two per 256-B bank row: xor (row >> 1) & 7) or 64 B (four per bank row: xor (row >> 2) & 3)
    auto row_swz = [](int row) { return CH == 8 ? (row >> 1) & 7 : (row >> 2) & 3; };
    using Desc = TileDesc<AJ, BJ>;
    // Tile order: groups of GM row tiles (GM <= 8); inside a group the row tile runs fastest, so the GM A tiles (GM x BM x K floats) stay in
    // the XCD's L2 while the group sweeps the columns and every B tile is fetched once per group, not once per row tile (the
    // Market-size search fetched its 33 MB gallery 27 times: PMC FETCH_SIZE 0.9 GB per launch against 40 MB of operands).
    // (GM row tiles of A must fit the L2 next to everything else: 1 MiB of them - 8 tiles at K = 96 .. 256, 4 at K = 512, the
    // plain row-major order from K = 2048 on, where the A tile itself is the large operand of a Swin fc2)
    const int a_tile_bytes = BM * p.K * 4;
    const int GM = a_tile_bytes >= (1 << 20) ? 1 : ((1 << 20) / a_tile_bytes > 8 ? 8 : (1 << 20) / a_tile_bytes);
    const int nmt = (p.M + BM - 1) / BM;
    auto describe = [&](int t, Desc& d) {
        const int group = t / (GM * nnt), first = group * GM;
        const int gm = nmt - first < GM ? nmt - first : GM;
        const int r = t - group * GM * nnt;
        const int ntile = r / gm, mtile = first + (r - ntile * gm);
        d.m_blk = mtile * BM;
        d.n_blk = ntile * BN;
        d.rows_a = p.M - d.m_blk < BM ? p.M - d.m_blk : BM;
        const int rows_b = p.N - d.n_blk < BN ? p.N - d.n_blk : BN;
        d.a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)p.A + (long long)d.m_blk * p.lda), 0,
                                                   (int)((long long)(d.rows_a - 1) * p.lda * 4 + (long long)p.K * 4), 0x00020000);
        d.b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (long long)d.n_blk * p.ldb), 0,
                                                   (int)((long long)(rows_b - 1) * p.ldb * 4 + (long long)p.K * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int row = (wave * AJ + j) * RPP + lane / CH;
            const int chunk = (lane % CH) ^ row_swz(row);
            d.a_voff[j] = row < d.rows_a ? (int)(((long long)row * p.lda + chunk * 4) * 4) : (int)0x7fffff00;
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = (wave * BJ + j) * RPP + lane / CH;
            const int chunk = (lane % CH) ^ row_swz(row);
            d.b_voff[j] = row < rows_b ? (int)(((long long)row * p.ldb + chunk * 4) * 4) : (int)0x7fffff00;
        }
    };
    auto stage = [&](const Desc& d, int kt, int slot) {
        char* As = lds + slot * STAGE;
        char* Bs = As + A_BYTES;
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(d.a_rs, LPTR(As + (wave * AJ + j) * 1024), 16, d.a_voff[j], kt * BK * 4, 0, 0);
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(d.b_rs, LPTR(Bs + (wave * BJ + j) * 1024), 16, d.b_voff[j], kt * BK * 4, 0, 0);
    };

    const int swz = row_swz(li);
    int a_rd[KK], b_rd[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const int pos = ((kk * 2 + lh) ^ swz) * 16;
        a_rd[kk] = (wm * 64 + li) * ROWB + pos;
        b_rd[kk] = A_BYTES + (wn * WN + li) * ROWB + pos;
    }
    f32x16 acc[TM][TN];
    auto mfma_tile = [&](int slot) {
        const char* base = lds + slot * STAGE;
        f32x4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[0][a] = *(const f32x4*)(base + a_rd[0] + a * 32 * ROWB);
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = *(const f32x4*)(base + b_rd[0] + b * 32 * ROWB);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if (kk < KK - 1) {
#pragma unroll
                for (int a = 0; a < TM; ++a) af[(kk + 1) & 1][a] = *(const f32x4*)(base + a_rd[kk + 1] + a * 32 * ROWB);
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[(kk + 1) & 1][b] = *(const f32x4*)(base + b_rd[kk + 1] + b * 32 * ROWB);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][a][e], bf[kk & 1][b][e], acc[a][b], 0, 0, 0);
        }
    };

    // ------------------------------------------------------------------ epilogues (registers -> memory, no LDS)
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    // Linear layers (everything but the ConvTranspose scatter): beside the other block's MFMAs every VALU instruction issues
    // at a fraction of its rate, so output and residual go through buffer instructions - per-lane byte offset fixed per column
    // block, row offset in the SGPR soffset, columns past N fall outside the descriptor and are dropped by the hardware
    // (full-height tiles only).  Per element: bias add, (erf-GELU,) residual add; no address arithmetic, no predicate.
    auto bias_lean = [&](const Desc& d, auto act_c, auto res_c) {
        constexpr bool ACT = decltype(act_c)::value, RES = decltype(res_c)::value;
        const int ldc = (int)p.ldc;
        const int recs = (int)(((long long)(d.rows_a - 1) * ldc + p.N) * 4);
        const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + (long long)d.m_blk * ldc), 0, recs, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)((RES ? p.residual : p.C) + (long long)d.m_blk * ldc), 0, recs, 0x00020000);
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = d.n_blk + wn * WN + b * 32 + li;
            const bool colok = col < p.N;
            const float sh = (colok && p.col_shift) ? p.col_shift[col] : 0.f;
            const int voff = colok ? ((wm * 64 + 4 * lh) * ldc + col) * 4 : 0x7fffff00;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                float res[16];
                if constexpr (RES) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        res[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, voff, (a * 32 + (e & 3) + 8 * (e >> 2)) * ldc * 4, 0));
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[a][b][e] + sh;
                    if constexpr (ACT) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));   // nn.GELU() (erf form)
                    if constexpr (RES) v += res[e];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rs, voff, (a * 32 + (e & 3) + 8 * (e >> 2)) * ldc * 4, 0);
                }
            }
        }
    };
    auto bias_general = [&](const Desc& d) {   // ConvTranspose parity scatter / huge row pitch: element by element
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = d.n_blk + wn * WN + b * 32 + li;
            const bool colok = col < p.N;
            const float sh = (colok && p.col_shift) ? p.col_shift[col] : 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = d.m_blk + wm * 64 + a * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
                    if (!colok || row >= p.M) continue;
                    long long orow = row;
                    if (p.scat_h > 0) {   // ConvTranspose2d(4,2,1) output parity: (img, j, i) -> (img, 2j+py, 2i+px)
                        const int hw = p.scat_h * p.scat_w;
                        const int img = row / hw, rem = row - img * hw;
                        const int j = rem / p.scat_w, i = rem - j * p.scat_w;
                        orow = ((long long)img * 2 * p.scat_h + 2 * j + p.scat_py) * (2 * p.scat_w) + 2 * i + p.scat_px;
                    }
                    const long long idx = orow * p.ldc + col;
                    float v = acc[a][b][e] + sh;
                    if (p.act == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    if (p.residual) v += p.residual[idx];
                    p.C[idx] = v;
                }
        }
    };
    // distance epilogue: metric chosen at compile time; full-height tiles store through the buffer path of the linear epilogue
    // (row offset in the SGPR operand, columns past N dropped by the descriptor), the ragged last row of tiles element by element
    auto dist_epilogue = [&](const Desc& d, auto metric_c) {
        constexpr int METRIC = decltype(metric_c)::value;
        const int ldc = (int)p.ldc;
        const bool full = d.rows_a == BM && (long long)BM * ldc * 4 < 0x7fffff00ll;
        const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + (long long)d.m_blk * ldc), 0,
                                                                              (int)(((long long)(BM - 1) * ldc + p.N) * 4), 0x00020000);
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = d.n_blk + wn * WN + b * 32 + li;
            const bool colok = col < p.N;
            float cq = (colok && p.col_sq) ? p.col_sq[col] : 0.f;
            if (METRIC == REID_METRIC_COS_HALF || METRIC == REID_METRIC_COS) cq = sqrtf(cq);
            const int voff = colok ? ((wm * 64 + 4 * lh) * ldc + col) * 4 : 0x7fffff00;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int row0 = d.m_blk + wm * 64 + a * 32 + 4 * lh;
                float rsq[16];   // squared norms of the tile's 16 rows of this lane half: four groups of four consecutive rows
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r4 = row0 + 8 * q;
                    if (p.row_sq && r4 + 3 < p.M && ((uintptr_t)(p.row_sq + r4) & 15) == 0) {
                        const f32x4 t = *(const f32x4*)(p.row_sq + r4);
                        rsq[4 * q] = t.x; rsq[4 * q + 1] = t.y; rsq[4 * q + 2] = t.z; rsq[4 * q + 3] = t.w;
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) rsq[4 * q + u] = (p.row_sq && r4 + u < p.M) ? p.row_sq[r4 + u] : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[a][b][e];
                    const float rs = rsq[e];
                    if (METRIC == REID_METRIC_L2) v = sqrtf(fmaxf(l2sqr_of(v, rs, cq), 1e-12f));
                    else if (METRIC == REID_METRIC_L2SQR) v = l2sqr_of(v, rs, cq);
                    else if (METRIC == REID_METRIC_COS_HALF) v = (1.0f - v / (sqrtf(rs) * cq)) / 2.0f;
                    else if (METRIC == REID_METRIC_COS) v = 1.0f - v / (sqrtf(rs) * cq);
                    if (full) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rs, voff, (a * 32 + (e & 3) + 8 * (e >> 2)) * ldc * 4, 0);
                    } else {
                        const int row = row0 + (e & 3) + 8 * (e >> 2);
                        if (colok && row < p.M) p.C[(long long)row * p.ldc + col] = v;
                    }
                }
            }
        }
    };
    // (the SGPR offset is not part of the descriptor's range check: a tile with fewer than BM rows takes the general loop)
    const bool lean_ok = EPI == E_BIAS && p.scat_h == 0 && (long long)BM * p.ldc * 4 < 0x7fffff00ll;

    // ------------------------------------------------------------------ one output tile per block
    // (Persistent blocks that treat the K-tiles of successive output tiles as one stream - the next tile's first operands in
    // flight under this tile's epilogue - were measured: stage-1 shapes unchanged, late stages slower from the static tile
    // assignment; 21.5 vs 19.2 ms for all linears of a 256-image pass.)
    Desc cur;
    int t;
    {   // XCD-aware, bijective block remap (blocks b and b+8 share an XCD): an XCD's blocks take consecutive tiles
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    describe(t, cur);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    stage(cur, 0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {       // unrolled by the two stages: LDS offsets become instruction immediates
            if (kt + u < nk) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the current K-tile have landed
                RAW_BARRIER();                                     // ... everyone's have, and everyone is done with the other stage
                if (kt + u + 1 < nk) stage(cur, kt + u + 1, u ^ 1);
                mfma_tile(u);
            }
        }
    }
    if constexpr (EPI == E_BIAS) {
        if (lean_ok && cur.rows_a == BM) {
            if (p.act == 1) {
                if (p.residual) bias_lean(cur, std::true_type{}, std::true_type{});
                else bias_lean(cur, std::true_type{}, std::false_type{});
            } else {
                if (p.residual) bias_lean(cur, std::false_type{}, std::true_type{});
                else bias_lean(cur, std::false_type{}, std::false_type{});
            }
        } else {
            bias_general(cur);
        }
    } else {
        switch (p.metric) {
            case REID_METRIC_L2: dist_epilogue(cur, std::integral_constant<int, REID_METRIC_L2>{}); break;
            case REID_METRIC_L2SQR: dist_epilogue(cur, std::integral_constant<int, REID_METRIC_L2SQR>{}); break;
            case REID_METRIC_COS_HALF: dist_epilogue(cur, std::integral_constant<int, REID_METRIC_COS_HALF>{}); break;
            case REID_METRIC_COS: dist_epilogue(cur, std::integral_constant<int, REID_METRIC_COS>{}); break;
            default: dist_epilogue(cur, std::integral_constant<int, -1>{}); break;   // plain dot products
        }
    }
#endif
}

template <int EPI>
void launch_epi(reid_ctx* ctx, const GemmParams& p) {
    const int nmt = (p.M + BM - 1) / BM;
    // Linear layers: the 64-wide tile everywhere - 149 VGPRs and 48 KB of LDS put THREE blocks on a CU (the 128-wide one: 223
    // VGPRs, two blocks), which hides more of the short K loops' load latency; measured on every Swin shape after the epilogue
    // rework (tools/bench_linear.py): 18.5 ms per pass against 18.9 (per-shape choice) and 19.4 (128-wide everywhere).
    // Distance matrices (long K, square-ish) keep the per-column cost model: 64-wide tiles run ~12 % below per column.
    const double cost64 = ((p.N + 63) / 64) * 64 * 1.12, cost128 = ((p.N + 127) / 128) * 128;
    if (EPI == E_BIAS || p.N <= 64 || cost64 < cost128) {
        hipLaunchKernelGGL((gemm_f32_dma_kernel<64, EPI>), dim3(nmt * ((p.N + 63) / 64)), dim3(256), 0, ctx->stream, p);
    } else if (EPI == E_DIST && ctx->f32_dist_bk16) {     // three blocks per CU (see the kernel's header)
        hipLaunchKernelGGL((gemm_f32_dma_kernel<128, EPI, 16>), dim3(nmt * ((p.N + 127) / 128)), dim3(256), 0, ctx->stream, p);
    } else {
        hipLaunchKernelGGL((gemm_f32_dma_kernel<128, EPI>), dim3(nmt * ((p.N + 127) / 128)), dim3(256), 0, ctx->stream, p);
    }
}

}  // namespace

bool gemm_f32_dma_supported(int amode, int epi, const GemmParams& p) {
    return amode == A_DENSE && (epi == E_BIAS || epi == E_DIST) && p.K % 32 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0 &&
           (long long)BM * p.lda * 4 < 0x7fff0000ll && (long long)128 * p.ldb * 4 < 0x7fff0000ll &&
           ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0;
}

int launch_gemm_f32_dma(reid_ctx* ctx, int epi, const GemmParams& p, int kind, double flops, double bytes) {
    prof_begin(ctx, kind, flops, bytes);
    if (epi == E_BIAS) launch_epi<E_BIAS>(ctx, p);
    else launch_epi<E_DIST>(ctx, p);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
