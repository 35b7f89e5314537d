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

// ------------------------------------------------------------------ epilogue (full tiles, no predicates)
// C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// v = acc * col_scale + col_shift (+ residual), ReLU on columns >= relu_from, per-(tile, column) sum / sumsq of the result.
// stat_lds: [2 (wm)][BN][2] floats of LDS that no wave reads as a tile any more (every wave is past the last barrier).
template <int BN>
__device__ __forceinline__ void conv_epilogue(const GemmParams& p, f32x16 (&acc)[2][BN / 64], float* stat_lds, int m_blk, int n_blk,
                                              int mtile, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins: device pass only
    constexpr int WN = BN / 2, TM = 2, TN = WN / 32;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const bool has_cs = p.col_scale != nullptr;
    // Output and residual through buffer instructions: per-lane byte offset fixed per column block, the row offset is a scalar
    // (SGPR soffset) - no per-element address arithmetic beside the other block's MFMAs, where a VALU instruction costs ~19 cycles.
    // Tiles are always full here (conv_f32_supported), so the unchecked scalar offset is safe.
    const int ldc = (int)p.ldc;
    const int recs = BM * ldc * 4;
    const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + (long long)m_blk * ldc), 0, recs, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)((p.residual ? p.residual : p.C) + (long long)m_blk * ldc), 0, recs, 0x00020000);
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int lcol = wn * WN + b * 32 + li;
        const int col = n_blk + lcol;
        const float cs = has_cs ? p.col_scale[col] : 1.f;
        const float sh = has_cs ? p.col_shift[col] : 0.f;
        const float lo = (p.relu && col >= p.relu_from) ? 0.f : -3.402823466e38f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int voff = ((wm * 64 + a * 32 + 4 * lh) * ldc + col) * 4;
            float res[16];
            if (p.residual) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    res[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, voff, ((e & 3) + 8 * (e >> 2)) * ldc * 4, 0));
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[a][b][e];
                if (has_cs) v = v * cs + sh;
                if (p.residual) v += res[e];
                v = fmaxf(v, lo);
                s1 += v;
                s2 += v * v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rs, voff, ((e & 3) + 8 * (e >> 2)) * ldc * 4, 0);
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
#endif
}

// Epilogue of the general variant (Swin patch merging, 8x8 alignment conv, ConvTranspose parities; swin_transformer.py:263-275,
// 405-412): bias, optional residual, optional parity scatter (img, j, i) -> (img, 2j+py, 2i+px); ragged M and N predicated.
template <int BN>
__device__ __forceinline__ void bias_epilogue(const GemmParams& p, f32x16 (&acc)[2][BN / 64], int m_blk, int n_blk, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WN = BN / 2, TM = 2, TN = WN / 32;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int ldc = (int)p.ldc;
    if (p.scat_h == 0 && 128ll * ldc * 4 < 0x7fffff00ll && p.M - m_blk >= 128) {
        // as gemm_f32_dma.hip's linear epilogue: buffer stores with the row offset in the SGPR operand (not range-checked: full
        // tiles only), columns past N dropped by the descriptor - per element one bias add (and one residual add)
        const int rows = 128;
        const int recs = (int)(((long long)(rows - 1) * ldc + p.N) * 4);
        const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + (long long)m_blk * ldc), 0, recs, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)((p.residual ? p.residual : p.C) + (long long)m_blk * ldc), 0, recs, 0x00020000);
        const bool has_res = p.residual != nullptr;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n_blk + wn * WN + b * 32 + li;
            const bool colok = col < p.N;
            const float sh = (colok && p.col_shift) ? p.col_shift[col] : 0.f;
            const int voff = colok ? ((wm * 64 + 4 * lh) * ldc + col) * 4 : 0x7fffff00;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                float res[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) res[e] = 0.f;
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        res[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, voff, (a * 32 + (e & 3) + 8 * (e >> 2)) * ldc * 4, 0));
                }
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[a][b][e] + sh + res[e]), c_rs, voff,
                                                          (a * 32 + (e & 3) + 8 * (e >> 2)) * ldc * 4, 0);
            }
        }
        return;
    }
    // scatter (ConvTranspose parity) and ragged-M tiles: element by element, the output row of each of the lane's 32 rows computed
    // once (divisions by multiply-high with host-made reciprocals)
    float sh[TN];
    int colv[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        colv[b] = n_blk + wn * WN + b * 32 + li;
        sh[b] = (colv[b] < p.N && p.col_shift) ? p.col_shift[colv[b]] : 0.f;
    }
    const unsigned hws = (unsigned)(p.scat_h * p.scat_w);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned row = (unsigned)(m_blk + wm * 64 + a * 32 + 4 * lh + (e & 3) + 8 * (e >> 2));
            if ((int)row >= p.M) continue;
            long long orow = row;
            if (p.scat_h > 0) {
                const unsigned im = __umulhi(row, p.scat_mhw), rem = row - im * hws;
                const unsigned j = __umulhi(rem, p.scat_mw), i = rem - j * (unsigned)p.scat_w;
                orow = ((long long)im * 2 * p.scat_h + 2 * j + p.scat_py) * (2 * p.scat_w) + 2 * i + p.scat_px;
            }
            const long long obase = orow * p.ldc;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                if (colv[b] >= p.N) continue;
                float v = acc[a][b][e] + sh[b];
                if (p.residual) v += p.residual[obase + colv[b]];
                p.C[obase + colv[b]] = v;
            }
        }
#endif
}

__device__ __forceinline__ unsigned long long stamp() {   // diagnostic builds only (cdna_hip_programming.md section 7, in-kernel stamps)
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <int BN, bool AFF, int FLAGS>
__global__ __launch_bounds__(256, 2) void conv_f32_kernel(const GemmParams p) {
    constexpr bool DIAG = (FLAGS & 1) != 0;   // s_memtime stamps (experiments)
    constexpr bool PRIO = (FLAGS & 2) != 0;   // raise the wave's priority while it stages the next tile
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
    // rows j*32 + lrow of the tile: Wo divides 32, so the four rows of a thread share ox and sit 32/Wo output rows apart
    int iy0[4], ix0;
    {
        const int rem = rem_blk + lrow;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        ix0 = ox * p.stride - p.pad_x;
#pragma unroll
        for (int j = 0; j < 4; ++j) iy0[j] = (oy + j * (32 / p.Wo)) * p.stride - p.pad_y;
    }
    const float* Aimg = (const float*)p.A + (long long)img * p.H * p.W * p.Cin + c4 * 4;
    const float* Bthr = p.B + (long long)(n_blk + lrow) * p.ldb + c4 * 4;
    const float* sc_ptr = AFF ? p.a_scale + (long long)img * p.Cin + c4 * 4 : nullptr;
    const float* sh_ptr = AFF ? p.a_shift + (long long)img * p.Cin + c4 * 4 : nullptr;
    const int cpt = p.Cin / BK;                     // K-tiles per tap
    const int wc = p.W * p.Cin;
    const int nk = p.R * p.S * cpt;

    // loader state (uniform): the K-tile the next issue_loads() fetches
    int l_r = 0, l_s = 0, l_cc = 0, l_k = 0;
    f32x4 ra[4], rb[BCH], rsc, rsh;
    unsigned okm = 0;
    auto issue_loads = [&]() {
        const int c0 = l_cc * BK;
        okm = 0;
        const int ix = ix0 + l_s;
        const bool okx = (unsigned)ix < (unsigned)p.W;
        const int xoff = min(max(ix, 0), p.W - 1) * p.Cin + c0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = iy0[j] + l_r;
            const bool ok = okx && (unsigned)iy < (unsigned)p.H;
            const int iyc = min(max(iy, 0), p.H - 1);
            ra[j] = *(const f32x4*)(Aimg + iyc * wc + xoff);
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
    unsigned long long dsum[5] = {0, 0, 0, 0, 0};
    for (int kt = 0; kt < nk; ++kt) {
        float* cur = lds + (kt & 1) * TILE;
        float* nxt = lds + ((kt + 1) & 1) * TILE;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
        if constexpr (DIAG) t0 = stamp();
        mfma_half(cur, 0);
        if constexpr (DIAG) t1 = stamp();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DIAG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t2 = stamp();
        }
        // The staging block is ~100 dependent VALU / LDS / VMEM instructions.  Beside the partner wave's back-to-back MFMAs (two
        // waves per SIMD, one per resident block) it took 3 000-3 600 cycles per K-tile (tools/diag_conv_f32.py): the partner's
        // MFMA needs ONE issue slot per 64 cycles, so this wave takes priority while it stages.
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
        if (kt + 1 < nk) {
            store_tile(nxt);                  // tile kt+1: its loads were issued one K-tile ago
            if (kt + 2 < nk) issue_loads();   // tile kt+2: lands behind the next 64 MFMAs
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DIAG) t3 = stamp();
        mfma_half(cur, 1);
        if constexpr (DIAG) t4 = stamp();
        __syncthreads();
        if constexpr (DIAG) {
            t5 = stamp();
            dsum[0] += t1 - t0; dsum[1] += t2 - t1; dsum[2] += t3 - t2; dsum[3] += t4 - t3; dsum[4] += t5 - t4;
        }
    }
    if constexpr (DIAG) {
        if (p.diag && blockIdx.x < 64 && lane == 0)
#pragma unroll
            for (int i = 0; i < 5; ++i) p.diag[(blockIdx.x * 8 + wave) * 5 + i] = dsum[i];
    }

    conv_epilogue<BN>(p, acc, (float*)lds, m_blk, n_blk, mtile, tid);
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (no input transform): the kernel every convolution of the fp32 forward runs on.
//
// Why: beside the OTHER wave's back-to-back v_mfma_f32_32x32x2_f32 a SIMD issues one VALU instruction per ~19 cycles
// instead of 3.3 (tools/bench_coissue.py: the fp32 MFMA holds the vector issue port for most of its 64 cycles; s_setprio
// changes nothing).  The register-staged kernel above spends ~100 VALU per K-tile and wave on addresses, the input
// affine, padding selects and LDS writes - 3 000-4 300 cycles per K-tile (tools/diag_conv_f32.py) in which the wave issues
// no MFMA, so the matrix pipe is only busy while its two waves happen to alternate.  Here a K-tile costs a wave ONE VALU
// instruction per A piece: buffer_load_dwordx4 ... lds with a per-lane byte offset fixed for the whole kernel (centre tap of
// the lane's output pixel), the tap / channel-chunk offset in the scalar offset, and padding as "offset out of range":
// g[j] holds the lane's (top, bottom, left, right) border flags in bits 31..28, sel the flags tap (r, s) violates,
// voffset = (g & sel) | voff  (v_and_or_b32) - any set bit pushes the offset past num_records = 2^28 and the DMA writes zeros.
// The LDS image is lane-linear (8 rows x 128 B per wave-instruction), conflict-free through an XOR swizzle of the source
// chunk and of the fragment read (chunk c of row r sits in slot c ^ ((r >> 1) & 7)).  Two stages: tile kt+1 lands while the
// 64 MFMAs of tile kt run; one s_waitcnt vmcnt(0) + one raw s_barrier per K-tile.
// The InstanceNorm / BatchNorm + ReLU that the register-staged kernel applied in its loader now happens where it is free:
// BatchNorm channels in the producing conv's epilogue (col_scale / col_shift, ReLU from column relu_from on), InstanceNorm
// channels in one in-place pass over half the tensor (in_apply_kernel, elementwise.hip).
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(uintptr_t)(p))
#define RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

template <int BN, int FLAGS>
__global__ __launch_bounds__(256, 2) void conv_f32_dma_kernel(const GemmParams p_in) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource builtins exist in the device pass only; the host pass needs just the stub
    GemmParams p = p_in;
    constexpr bool DIAG = (FLAGS & 1) != 0;
    constexpr bool GEN = (FLAGS & 4) != 0;          // general geometry: tiles may straddle images, ragged M / N, bias epilogue
    constexpr int WN = BN / 2, TM = 2, TN = WN / 32;
    constexpr int ROWB = BK * 4;                    // 128-byte LDS rows
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int AJ = BM / 8 / 4;                  // A pieces (8 rows each) per wave and K-tile
    constexpr int BJ = BN / 8 / 4;
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int nnt = (p.N + BN - 1) / BN;
    int mtile, ntile, ksplit = 0;
    {
        const int nwg = gridDim.x;
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        if constexpr (!GEN) {
            if (p.split_k > 1) {   // SK blocks per output tile, each summing nk / SK of the K-tiles (small launches)
                ksplit = L % p.split_k;
                L /= p.split_k;
            }
        }
        if constexpr (GEN) {
            if (p.par4) {   // four ConvTranspose parities in one launch: copy q of the tile grid = parity (q >> 1, q & 1)
                const int tiles = nwg >> 2, par = L / tiles;
                L -= par * tiles;
                p.B += par * p.par_stride;
                p.pad_y = 1 - (par >> 1); p.pad_x = 1 - (par & 1);
                p.scat_py = par >> 1; p.scat_px = par & 1;
            }
        }
        mtile = L / nnt;
        ntile = L - mtile * nnt;
    }
    const int m_blk = mtile * BM, n_blk = ntile * BN;
    const int hw = p.Ho * p.Wo;
    const int img = m_blk / hw;                     // first image of the tile (the only one unless GEN)
    const int rem_blk = m_blk - img * hw;

    // ---- DMA descriptors
    // A: base = first byte the top-left tap of pixel (0, 0) would read (before the image when pad > 0: never dereferenced,
    // those lanes are flagged); num_records 2^28 > image + tap offsets, flagged offsets are >= 2^28
    const char* a_base = (const char*)p.A + ((long long)img * p.H * p.W - (p.pad_y * p.W + p.pad_x)) * (long long)p.Cin * 4;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, 1 << 28, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, GEN ? 0x7ffffe00 : 0x7fffffff, 0x00020000);
    int a_voff[AJ];
    unsigned a_flag[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = (wave * AJ + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int rem = rem_blk + row, dimg = 0;
        if constexpr (GEN) {                                    // the tile may run over into the next images
            dimg = rem / hw;
            rem -= dimg * hw;
        }
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int iy = oy * p.stride, ix = ox * p.stride;       // pixel of tap (pad_y, pad_x): always inside the image
        a_voff[j] = (((dimg * p.H + iy) * p.W + ix) * p.Cin + chunk * 4) * 4;
        unsigned f = 0;
        if (iy - p.pad_y < 0) f |= 1u << 31;                    // tap row 0 is above the image
        if (iy - p.pad_y + p.R - 1 >= p.H) f |= 1u << 30;       // tap row R-1 is below it
        if (ix - p.pad_x < 0) f |= 1u << 29;
        if (ix - p.pad_x + p.S - 1 >= p.W) f |= 1u << 28;
        a_flag[j] = f;
        if constexpr (GEN) {
            if (m_blk + row >= p.M) a_voff[j] = 0x7fffff00;    // rows past M: out of range for every tap (num_records = 2^28)
        }
    }
    int b_voff[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = (wave * BJ + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        b_voff[j] = (int)(((long long)(n_blk + row) * p.ldb + chunk * 4) * 4);
        if constexpr (GEN) {
            if (n_blk + row >= p.N) b_voff[j] = 0x7fffff00;    // weight rows past N (beyond b_rs' num_records)
        }
    }
    const int cpt = p.Cin / BK;
    const int SK = (!GEN && p.split_k > 1) ? p.split_k : 1;
    const int nk = p.R * p.S * cpt / SK;         // K-tiles of this block: [ksplit * nk, (ksplit + 1) * nk)
    const int wc4 = p.W * p.Cin * 4;

    int l_r, l_s, l_cc, l_k;                     // uniform: the K-tile the next stage() fetches
    {
        const int kt0 = ksplit * nk, tap = kt0 / cpt;
        l_cc = kt0 - tap * cpt;
        l_r = tap / p.S;
        l_s = tap - l_r * p.S;
        l_k = kt0 * BK;
    }
    auto stage = [&](int slot) {
        char* As = lds + slot * STAGE;
        char* Bs = As + A_BYTES;
        const int soff = l_r * wc4 + (l_s * p.Cin + l_cc * BK) * 4;
        const unsigned sel = (l_r == 0 ? 1u << 31 : 0u) | (l_r == p.R - 1 ? 1u << 30 : 0u) | (l_s == 0 ? 1u << 29 : 0u) |
                             (l_s == p.S - 1 ? 1u << 28 : 0u);
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, LPTR(As + (wave * AJ + j) * 1024), 16, (int)((a_flag[j] & sel) | (unsigned)a_voff[j]),
                                                     soff, 0, 0);
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, LPTR(Bs + (wave * BJ + j) * 1024), 16, b_voff[j], l_k * 4, 0, 0);
        l_k += BK;
        if (++l_cc == cpt) {
            l_cc = 0;
            if (++l_s == p.S) { l_s = 0; ++l_r; }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // fragment reads: lane (li, lh) takes chunk kk*2 + lh of row li (k = 8 kk + 4 lh ..+3), slot = chunk ^ ((row >> 1) & 7)
    const int swz = (li >> 1) & 7;
    int a_rd[4], b_rd[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int pos = ((kk * 2 + lh) ^ swz) * 16;
        a_rd[kk] = (wm * 64 + li) * ROWB + pos;
        b_rd[kk] = A_BYTES + (wn * WN + li) * ROWB + pos;
    }
    // fragments of k-step kk+1 are fetched before the 16 MFMAs of k-step kk are issued (two register sets)
    auto mfma_tile = [&](int slot) {
        const char* base = lds + slot * STAGE;
        f32x4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[0][a] = *(const f32x4*)(base + a_rd[0] + a * 32 * ROWB);
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = *(const f32x4*)(base + b_rd[0] + b * 32 * ROWB);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) {
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

    unsigned long long dsum[5] = {0, 0, 0, 0, 0};
    stage(0);
    for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {       // unrolled by the two stages: LDS offsets become instruction immediates
            if (kt + u < nk) {
                unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
                if constexpr (DIAG) t0 = stamp();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt+u have landed
                if constexpr (DIAG) t1 = stamp();
                RAW_BARRIER();                                     // ... everyone's have, and everyone is done reading tile kt+u-1
                if constexpr (DIAG) t2 = stamp();
                if (kt + u + 1 < nk) stage(u ^ 1);
                if constexpr (DIAG) t3 = stamp();
                mfma_tile(u);
                if constexpr (DIAG) {
                    t4 = stamp();
                    dsum[0] += t1 - t0; dsum[1] += t2 - t1; dsum[2] += t3 - t2; dsum[3] += t4 - t3;
                }
            }
        }
    }
    if constexpr (DIAG) {
        if (p.diag && blockIdx.x < 64 && lane == 0)
#pragma unroll
            for (int i = 0; i < 5; ++i) p.diag[(blockIdx.x * 8 + wave) * 5 + i] = dsum[i];
    }
    if constexpr (GEN) {
        bias_epilogue<BN>(p, acc, m_blk, n_blk, tid);
    } else {
        __syncthreads();   // the epilogue reuses the tiles' LDS for the column statistics
        if (SK > 1) {
            // split-K (as conv3x3_f16.hip): fp32 partial tiles as device-scope (sc1) stores, an arrival counter per output tile,
            // the block that arrives last sums the partials in split order (deterministic) and runs the epilogue.  No L2
            // write-back / invalidate fence: the stores are written through, the loads bypass this XCD's L2.
            constexpr int PART = BM * BN;
            const int tile_id = mtile * nnt + ntile;
            float* part = p.splitk_ws + (long long)tile_id * SK * PART;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        __hip_atomic_store(part + (long long)ksplit * PART + ((a * TN + b) * 16 + e) * 256 + tid, acc[a][b][e], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* flag = (int*)lds;
            if (tid == 0) {
                const int old = __hip_atomic_fetch_add(p.splitk_cnt + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = old == SK - 1;
                if (old == SK - 1) __hip_atomic_store(p.splitk_cnt + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (!*flag) return;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
            for (int sidx = 0; sidx < SK; ++sidx)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            acc[a][b][e] += __hip_atomic_load(part + (long long)sidx * PART + ((a * TN + b) * 16 + e) * 256 + tid, __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();   // the flag word is about to become statistics scratch
        }
        conv_epilogue<BN>(p, acc, (float*)lds, m_blk, n_blk, mtile, tid);
    }
#endif
}

bool conv_f32_dma_supported(const GemmParams& p) {
    return (p.R == 1 || p.R == 3) && p.S == p.R && p.pad_y == (p.R - 1) / 2 && p.pad_x == p.pad_y &&
           ((long long)p.H * p.W + 4ll * p.W + 4) * p.Cin * 4 < (1ll << 28) && (long long)p.N * p.ldb * 4 < 0x7fffffffll;
}

template <int BN>
void launch_bn(reid_ctx* ctx, const GemmParams& p0) {
    GemmParams p = p0;
    int grid = (p.M / BM) * (p.N / BN);
    // a launch that leaves CUs idle or alone (a tracking frame): 2 or 4 blocks per output tile, each with a share of the K-tiles
    if (BN == 64 && ctx->f32_split_k && !p.a_scale && ctx->f32_conv != 2 && conv_f32_dma_supported(p) && !p.diag) {
        const int nk = p.R * p.S * (p.Cin / BK);
        int sk = 1;
        // (round 6: up to 768 blocks - three fit a CU; at 512, a frame of 33 crops ran its two strided 3x3 convolutions as 264 unsplit blocks,
        // 42-45 us each against 29-30 us for the 512 half-length blocks of 32 crops.  Launches of 256 and 512 tiles split as before.)
        while (sk < 4 && grid * sk * 2 <= 768 && nk % (sk * 2) == 0 && nk / (sk * 2) >= 8) sk *= 2;
        // between the powers of two: three ways where two leave a third of the block slots empty, and a split for launches of
        // 257 .. 511 tiles, which used to run unsplit at full K length (a frame of 33 crops took 2.12 ms against 1.29 ms for 32).
        // Launches of exactly 256 or 512 tiles (passes of 64 crops) keep the form they had.
        // (only K loops that stay long: layer 1's 18 K-tiles split two ways were slower than unsplit)
        if (sk == 2 && grid * 3 <= 512 && nk % 3 == 0 && nk / 3 >= 16) sk = 3;
        else if (sk == 1 && grid > 256 && grid < 512) {
            if (nk % 3 == 0 && nk / 3 >= 16 && grid * 3 <= 1024) sk = 3;
            else if (nk % 2 == 0 && nk / 2 >= 16) sk = 2;
        }
        if (sk > 1 && grid <= 1024) {
            float* ws;
            int* cnt;
            const bool fresh = ctx->ws.find("conv32.splitk_cnt") == ctx->ws.end();
            if (ctx_ws(ctx, "conv32.splitk_ws", (size_t)grid * sk * BM * BN * sizeof(float), (void**)&ws) == REID_OK &&
                ctx_ws(ctx, "conv32.splitk_cnt", 1024 * sizeof(int), (void**)&cnt) == REID_OK) {
                if (fresh) (void)hipMemsetAsync(cnt, 0, 1024 * sizeof(int), ctx->stream);
                p.split_k = sk; p.splitk_ws = ws; p.splitk_cnt = cnt;
                grid *= sk;
            }
        }
    }
#ifdef REID_EXPERIMENTS
    const bool staged_everywhere = ctx->f32_conv == 2;     // switch f32_conv = 2: the register-staged kernel everywhere (A/B)
#else
    const bool staged_everywhere = false;
#endif
    if (!p.a_scale && !staged_everywhere && conv_f32_dma_supported(p)) {
        if (p.diag) hipLaunchKernelGGL((conv_f32_dma_kernel<BN, 1>), dim3(grid), dim3(256), 0, ctx->stream, p);
        else hipLaunchKernelGGL((conv_f32_dma_kernel<BN, 0>), dim3(grid), dim3(256), 0, ctx->stream, p);
        return;
    }
#ifdef REID_EXPERIMENTS
    const bool prio = false;   // s_setprio around the staging block: measured, no gain (the MFMA holds the issue port)
#define CONV_F32_LAUNCH(AFF, FL) hipLaunchKernelGGL((conv_f32_kernel<BN, AFF, FL>), dim3(grid), dim3(256), 0, ctx->stream, p)
    if (p.diag) {   // experiments: the same kernel with s_memtime stamps around the segments of a K-tile
        if (p.a_scale) { if (prio) CONV_F32_LAUNCH(true, 3); else CONV_F32_LAUNCH(true, 1); }
        else { if (prio) CONV_F32_LAUNCH(false, 3); else CONV_F32_LAUNCH(false, 1); }
        return;
    }
    if (p.a_scale) { if (prio) CONV_F32_LAUNCH(true, 2); else CONV_F32_LAUNCH(true, 0); }
    else { if (prio) CONV_F32_LAUNCH(false, 2); else CONV_F32_LAUNCH(false, 0); }
#undef CONV_F32_LAUNCH
#else
    // the register-staged kernel of round 1 (switch f32_conv = 2, or a transform in the loader): 16 builds, in libraries made with
    // -DREID_EXPERIMENTS only; conv_f32_supported() keeps such convolutions away from here in the product library
    (void)grid;
#endif
}

}  // namespace

bool conv_f32_supported(const GemmParams& p) {
    return p.Cin % BK == 0 && p.K == p.R * p.S * p.Cin && p.M % BM == 0 && (p.Ho * p.Wo) % BM == 0 && p.N % 64 == 0 &&
           p.Wo >= 1 && p.Wo <= 32 && 32 % p.Wo == 0 &&
           p.ldb % 4 == 0 && p.ldc == p.N && (p.a_scale == nullptr) == (p.a_shift == nullptr) &&
           (p.col_scale == nullptr) == (p.col_shift == nullptr) && (long long)p.H * p.W * p.Cin < (1ll << 31)
#ifndef REID_EXPERIMENTS
           && !p.a_scale && conv_f32_dma_supported(p)      // (the register-staged kernel is not in the product library)
#endif
        ;
}

bool conv_f32_general_supported(const GemmParams& p) {
    return p.Cin % BK == 0 && p.K == p.R * p.S * p.Cin && p.R == p.S && p.R >= 1 && p.R <= 8 && p.pad_y >= 0 && p.pad_y <= 1 &&
           p.pad_x >= 0 && p.pad_x <= 1 && p.ldb % 4 == 0 && !p.a_scale && !p.stats && !p.col_scale && !p.relu && p.act == 0 &&
           ((long long)5 * p.H * p.W + 16ll * p.W + 16) * p.Cin * 4 < (1ll << 28) && (long long)p.N * p.ldb * 4 < 0x7ffffe00ll &&
           (long long)p.Ho * p.Wo * 4 >= BM;   // a 128-row tile touches at most 5 images
}

// Swin's convolutions with bias (patch merging 2x2 s2, alignment 8x8 s8, ConvTranspose parities 2x2 s1 with one-sided padding)
int launch_conv_f32_general(reid_ctx* ctx, const GemmParams& p, int kind, double flops, double bytes) {
    ARG_CHECK(conv_f32_general_supported(p));
    prof_begin(ctx, kind, flops, bytes);
    const int nmt = (p.M + BM - 1) / BM;
    const int copies = p.par4 ? 4 : 1;   // the four ConvTranspose parities as one grid
    // the 64-wide tile (three blocks per CU) on every shape: Swin fp32 8.65 -> 8.86 k img/s against the per-column cost model
    hipLaunchKernelGGL((conv_f32_dma_kernel<64, 4>), dim3(nmt * ((p.N + 63) / 64) * copies), dim3(256), 0, ctx->stream, p);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_conv_f32(reid_ctx* ctx, const GemmParams& p, int kind, double flops, double bytes) {
    ARG_CHECK(conv_f32_supported(p));
    prof_begin(ctx, kind, flops, bytes);
    // 128-wide tiles when they fill the chip; a tracking-sized batch (30 crops: 30 M-tiles) gets twice the blocks from 64-wide tiles,
    // each with half the K-loop time
    if (p.N % 128 == 0 && (long long)(p.M / BM) * (p.N / 128) >= 256) launch_bn<128>(ctx, p);
    else launch_bn<64>(ctx, p);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
