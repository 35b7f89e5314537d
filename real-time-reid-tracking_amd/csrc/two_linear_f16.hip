// Two Linear layers in one launch, fp32-class arithmetic (reid_ctx_set_precision(ctx, 2)):
//     out = res + W2 . act(W1 . x + b1) + b2            x: [T][C] as [xh | xl'] f16, W1: [HID][C], W2: [C][HID], fp32 res / out
// the Swin block's MLP (fc1 -> GELU -> fc2 + x, swin_transformer.py:23-39) and its to_out -> post_proj pair (+ x, :66-82,191-232).
// x comes packed ([T][2C] f16) or as the fp32 residual stream with LayerNorm 2 applied in the kernel's prologue (:228).  At the end of
// the file: LayerNorm 1 + to_qkv as one launch on the same register-resident tokens (ln_linear_f16x3_kernel).
//
// Why: in stages 1-2 (C = 96 / 192) these pairs are bound by the HIDDEN tensor's trip through HBM, not by arithmetic - the MLP of
// stage 1 writes and re-reads [T][4C] values (2.5 GB per 256 images) around 0.36 TFLOP.  Here the hidden values never leave the
// register file:
//   * a wave owns 32 tokens for the whole kernel; their [xh | xl'] rows sit in VGPRs as MFMA B fragments (C/2 registers);
//   * the hidden layer is walked in groups of 32 units.  GEMM 1 runs TRANSPOSED - H^T[hid][tok] = W1[hid][:] . x^T - so the
//     accumulator of a 32x32 tile holds, per lane, ONE token (column = lane & 31) and 16 hidden units; bias, GELU and the
//     [hh | hl'] split happen on those registers and the two half8 they pack into ARE the A fragments of GEMM 2
//     (out[tok][n] += H[tok][hid] . W2[n][hid]) - with the hidden units of a k-step in accumulator order
//     {0-3, 8-11} + 4 (lane >> 5) + 16 s, which the W2 image is laid out to match (two_linear_w2_tiles_kernel);
//   * the three products of the fp32-class form (conv3x3_f16.hip): x.w = xh.wh + (xl'.wh + xh.wl') 2^-11, accumulated as
//     [xh | xl' | xh] . [wh 2^11 | wh | wl'] and scaled by 2^-11 in the epilogue - for both layers;
//   * weights stream through LDS once per block as 2-KB tiles [32 rows][32 k] that the load-time kernels below write in
//     exactly the (swizzled) order the LDS image has, so a step's weights are one linear global -> LDS DMA copy.
// Every token's arithmetic is the same instruction sequence whichever row of a tile it lands on (no per-row special cases, f16
// conversions through cvt_f16_rn): images stay independent of their position in the batch.
#include "reid_internal.h"
#include <type_traits>

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#include "lin_math.h"

namespace {

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(uintptr_t)(p))
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

// chunk (8 f16) c of row r of a [32][32] tile sits at position c ^ ((r >> 2) & 3): the 16 rows of a ds_read_b128 lane group
// cover the 16 slots of a 256-byte bank row (the 64-byte-row scheme of gemm_f16.hip)
__device__ __forceinline__ int tile_pos(int row, int k) { return row * 32 + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7); }

__device__ __forceinline__ f16 split_part(float v, int part) {
    const f16 wh = (f16)v;
    return part == 0 ? (f16)((float)wh * 2048.0f) : part == 1 ? wh : (f16)((v - (float)wh) * 2048.0f);
}

// W1 fp32 [HID][C] -> tiles [g][kt][ih][32 hidden][32 virtual k] over the virtual K = [wh 2^11 | wh | wl'] (3 C columns)
__global__ __launch_bounds__(256) void two_linear_w1_tiles_kernel(const float* __restrict__ w, int hid, int C, int HG, f16* __restrict__ out) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= (long long)hid * 3 * C) return;
    const int h = (int)(i / (3 * C)), vk = (int)(i - (long long)h * 3 * C);
    const int part = vk / C, c = vk - part * C;
    const int g = h / (32 * HG), ih = (h >> 5) % HG, row = h & 31, kt = vk >> 5;
    const long long tile = ((long long)g * (3 * C / 32) + kt) * HG + ih;
    out[tile * 1024 + tile_pos(row, vk & 31)] = split_part(w[(long long)h * C + c], part);
}

// W2 fp32 [C][HID] -> tiles [g][j][part][ih][32 outputs][32 hidden], hidden unit u (0..31) of a tile at k position
// 16 s + 8 lh + i with u = 16 s + (i & 3) + 8 (i >> 2) + 4 lh: the order GEMM 1's accumulator registers hold them in
__global__ __launch_bounds__(256) void two_linear_w2_tiles_kernel(const float* __restrict__ w, int hid, int C, int HG, f16* __restrict__ out) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= (long long)C * 3 * hid) return;
    const int n = (int)(i / (3 * hid)), rem = (int)(i - (long long)n * 3 * hid);
    const int part = rem / hid, h = rem - part * hid;
    const int g = h / (32 * HG), ih = (h >> 5) % HG, u = h & 31;
    const int s = u >> 4, r = u & 15;
    const int kpos = 16 * s + 8 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
    const int j = n >> 5, row = n & 31;
    const long long tile = (((long long)g * (C / 32) + j) * 3 + part) * HG + ih;
    out[tile * 1024 + tile_pos(row, kpos)] = split_part(w[(long long)n * hid + h], part);
}

// LayerNorm (swin_transformer.py:195,228; eps 1e-5) of one fp32 token row, then the [yh | yl'] split, in registers, as the MFMA
// fragments of that token: the lane holds columns 16 r + 8 lh .. + 7 (half the row), its partner lane ^ 32 the rest; two-pass
// statistics as layernorm_v4_kernel's.  lns: [gamma | beta] in LDS.  Returns the running range-guard word.
template <int C>
__device__ __forceinline__ unsigned token_fragments_ln(const float* __restrict__ row, const float* lns, int lh, half8 (&xf)[2 * C / 16], unsigned xmax) {
    constexpr int KH = C / 16;
    const float* xrow = row + 8 * lh;
    f32x4 xv[2 * KH];
#pragma unroll
    for (int r = 0; r < KH; ++r) {
        xv[2 * r] = *(const f32x4*)(xrow + 16 * r);
        xv[2 * r + 1] = *(const f32x4*)(xrow + 16 * r + 4);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * KH; ++i) sum += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
    sum += __shfl_xor(sum, 32);
    const float mean = sum / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * KH; ++i) {
        const f32x4 d = xv[i] - mean;
        q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    q += __shfl_xor(q, 32);
    const float rstd = 1.0f / sqrtf(q / C + 1e-5f);
#pragma unroll
    for (int r = 0; r < KH; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 gg = *(const f32x4*)&lns[16 * r + 8 * lh + 4 * h], bb = *(const f32x4*)&lns[C + 16 * r + 8 * lh + 4 * h];
            const f32x4 y = (xv[2 * r + h] - mean) * rstd * gg + bb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xmax = range_acc(xmax, y[i]);
                const f16 hv = cvt_f16_rn(y[i]);
                xf[r][4 * h + i] = hv;
                xf[KH + r][4 * h + i] = cvt_f16_rn((y[i] - (float)hv) * 2048.0f);
            }
        }
    }
    return xmax;
}

struct TwoLinearParams {
    const f16* A;        // [T][2C]: [xh | xl'] - or null: LayerNorm(X32) computed here
    const float* X32;    // [T][C] fp32 (LN builds)
    const float* ln_g;   // [C]
    const float* ln_b;
    long long T;
    const f16* W1t;      // two_linear_w1_tiles_kernel
    const f16* W2t;      // two_linear_w2_tiles_kernel
    const float* b1;     // [HID]
    const float* b2;     // [C]
    const float* res;    // [T][C] fp32 (may alias out)
    float* out;          // [T][C] fp32
    int hid;
    int* fault;
    int ablate;          // experiments (timing only): 1 = no weight refills after the first two steps, 2 = no block barriers
};

// One block = NW waves x 32 tokens.  STEP it (0 .. NG + 1) of a wave, NG = HID / 32 hidden groups:
//     G1(it)    : hnext = W1[group it] . x^T                        18 C/96 MFMAs   (it < NG)
//     E(it - 1) : bias, activation, [hh | hl'] of group it - 1      VALU            (1 <= it <= NG)
//     G2(it - 2): oacc += [hh | hl' | hh] . W2[group it - 2]        18 C/96 MFMAs   (it >= 2)
// written as ONE instruction stream of 16 chunks - chunk e = the VALU work of accumulator register e + its share of the two
// GEMMs' MFMAs and the fragment reads of two chunks ahead - with a scheduling barrier between chunks: the matrix pipe works on
// groups it and it - 2 while the VALU runs the GELU of group it - 1.  (Phase by phase - GEMM 1, barrier, GELU, GEMM 2, barrier -
// every wave of the block does the same kind of work at the same time and nothing overlaps: 708 us per stage-1 MLP of 256 images
// on random operands (tools/two_linear_check.py) against 610-630 in this order; 480-490 us inside the Swin pass, whose operands
// let the chip hold a higher clock.)
// The weights of a step ([W1 group it | W2 group it - 2], 2-KB tiles as the load-time kernels wrote them) are one linear
// global -> LDS DMA copy into one of two slots; one block barrier per step.
template <int C, int NW, bool ACT, int AHEAD, int TT, bool LN>
__global__ __launch_bounds__(NW * 64, (TT == 1 && C <= 96) ? 2 : 1) void two_linear_f16x3_kernel(const TwoLinearParams p) {
    constexpr int NT = 3 * C / 32;           // tiles per group and GEMM: GEMM 1 K-tiles of 32 / GEMM 2 (output tile, part)
    constexpr int KS1 = 3 * C / 16;          // GEMM 1 k-steps over the virtual K
    constexpr int KR = 2 * C / 16;           // x fragments held (real K)
    constexpr int NJ = C / 32;               // output column tiles
    constexpr int NM2 = 2 * 3 * NJ;          // GEMM 2 MFMAs per group: (k-step, part, output tile)
    constexpr int HB = NT * 2048;            // bytes of one group of one weight matrix
    constexpr int SB = 2 * HB;               // ... of a step
    constexpr int PI = SB / 1024;            // DMA wave-instructions per step
    constexpr int IPW = (PI + NW - 1) / NW;  // ... per wave (the surplus repeats earlier pieces: same bytes, same place)
    constexpr int NS = 2;                    // steps resident: the one being read, the one landing
    constexpr int BT = NW * 32 * TT;         // tokens per block (TT 32-token tiles per wave)
    static_assert(KS1 == NM2, "the two GEMMs of a group are the same number of MFMAs");
    static_assert(NS * SB + 16 * C <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char lds[NS * SB];
    __shared__ __attribute__((aligned(16))) float b1s[4 * C];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const long long tok0 = (long long)blockIdx.x * BT;
    const int NG = p.hid / 32;
    const int last = NG + 1;                 // steps 0 .. NG + 1

    __shared__ __attribute__((aligned(16))) float lns[LN ? 2 * C : 4];
    for (int i = tid; i < p.hid; i += NW * 64) b1s[i] = p.b1[i];
    if constexpr (LN) {
        for (int i = tid; i < C; i += NW * 64) {
            lns[i] = p.ln_g[i];
            lns[C + i] = p.ln_b[i];
        }
    }
    __syncthreads();                         // (before any DMA is in flight: the only full drain of the kernel)

    // ---- this lane's token row as B fragments: k-step r covers columns 16 r .. 16 r + 15 of [xh | xl'], 8 per lane half
    half8 xf[TT][KR];
    unsigned xmax = 0u;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        long long tok = tok0 + (wave * TT + t) * 32 + li;
        if (tok >= p.T) tok = p.T - 1;       // a ragged last block: a valid row, its results fall outside the store descriptor
        if constexpr (LN) xmax = token_fragments_ln<C>(p.X32 + tok * C, lns, lh, xf[t], xmax);
        else {
            const f16* arow = p.A + tok * (2 * C) + 8 * lh;
#pragma unroll
            for (int r = 0; r < KR; ++r) xf[t][r] = *(const half8*)(arow + 16 * r);
        }
    }

    auto issue_step = [&](int it, int slot) {   // absent halves (it >= NG, it < 2) copy a neighbouring group: never read
        const int g1 = it < NG ? it : NG - 1, g2 = it < 2 ? 0 : (it - 2 < NG ? it - 2 : NG - 1);
        const char* s1 = (const char*)p.W1t + (long long)g1 * HB;
        const char* s2 = (const char*)p.W2t + (long long)g2 * HB - HB;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            int q = wave + NW * j;
            if (q >= PI) q -= PI;
            const char* src = (q * 1024 < HB ? s1 : s2) + q * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(lds + slot * SB + q * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < 2; ++s) issue_step(s, s);          // NG >= 1: at least three steps

    f32x16 oacc[TT][NJ];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[t][j][e] = 0.f;

    // fragment byte offset inside a tile for k-step kk (0, 1) of its 32 k
    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = li * 64 + (((kk * 2 + lh) ^ (li >> 2)) & 3) * 16;

    int slot_c = 0;                          // slot being read; the next DMA goes to the other one
    float vmax = 0.f;
    f32x16 hcur[TT], hnext[TT];              // GEMM 1 accumulators: the group in the epilogue / the group being summed
    half8 Hh[TT][2], Hl[TT][2];              // the group GEMM 2 reads
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) hcur[t][e] = 0.f;

    // MFMAs of chunk c: GEMM indices [mlo(c), mlo(c + 1)) - KS1 over 16 chunks
    auto mlo = [](int c) { return c * KS1 / 16; };
    constexpr int PF = AHEAD * KS1 / 16;     // fragments read before a step's first chunk
    // fragment of MFMA m of each GEMM in ring slot `slot`: GEMM 1 k-step m; GEMM 2 m = (s * 3 + part) * NJ + j
    auto read1 = [&](int slot, int m) { return *(const half8*)(lds + slot * SB + (m >> 1) * 2048 + foff[m & 1]); };
    auto read2 = [&](int slot, int m) {
        const int j = m % NJ, sp = m / NJ, part = sp % 3, s = sp / 3;
        return *(const half8*)(lds + slot * SB + HB + (j * 3 + part) * 2048 + foff[s]);
    };
    auto step = [&](int it, auto g1_c, auto e_c, auto g2_c) {
        constexpr bool G1 = decltype(g1_c)::value, E = decltype(e_c)::value, G2 = decltype(g2_c)::value;
        half8 f1[KS1], f2[NM2];
        f32x4 bq[4];
        half8 nHh[TT][2], nHl[TT][2];
        // own pieces of step it landed -> barrier (everyone's did, and step it - 1 has been read) -> step it + 1 into the other slot
        if (it == 0) WAIT_VMCNT(IPW);        // (steps 0 and 1 were issued before the loop)
        else WAIT_VMCNT(0);
        if (!(p.ablate & 2)) RAW_BARRIER();
        if (it >= 1 && it + 1 <= last && !(p.ablate & 1)) issue_step(it + 1, slot_c ^ 1);
#pragma unroll
        for (int m = 0; m < PF; ++m) {
            if constexpr (G1) f1[m] = read1(slot_c, m);
            if constexpr (G2) f2[m] = read2(slot_c, m);
        }
        if constexpr (E) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *(const f32x4*)&b1s[(it - 1) * 32 + 8 * q + 4 * lh];
        }
        if constexpr (G1) {
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) hnext[t][e] = 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c + AHEAD < 16) {
#pragma unroll
                for (int m = mlo(c + AHEAD); m < mlo(c + AHEAD + 1); ++m) {
                    if constexpr (G1) f1[m] = read1(slot_c, m);
                    if constexpr (G2) f2[m] = read2(slot_c, m);
                }
            }
#pragma unroll
            for (int m = mlo(c); m < mlo(c + 1); ++m) {
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    if constexpr (G1) hnext[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1[m], xf[t][m % KR], hnext[t], 0, 0, 0);
                    if constexpr (G2) {
                        const int j = m % NJ, sp = m / NJ, part = sp % 3, s = sp / 3;
                        oacc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(part == 1 ? Hl[t][s] : Hh[t][s], f2[m], oacc[t][j], 0, 0, 0);
                    }
                }
            }
            if constexpr (E) {   // register c of lane half lh is hidden unit (c & 3) + 8 (c >> 2) + 4 lh of the group
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    float v = fmaf(hcur[t][c], 1.0f / 2048.0f, bq[c >> 2][c & 3]);
                    if constexpr (ACT) v = gelu_f16_storage(v);
                    vmax = fmaxf(vmax, fabsf(v));
                    const f16 hv = cvt_f16_rn(v);
                    nHh[t][c >> 3][c & 7] = hv;
                    nHl[t][c >> 3][c & 7] = cvt_f16_rn((v - (float)hv) * 2048.0f);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if constexpr (E) {
#pragma unroll
                for (int s = 0; s < 2; ++s) { Hh[t][s] = nHh[t][s]; Hl[t][s] = nHl[t][s]; }
            }
            if constexpr (G1) hcur[t] = hnext[t];
        }
        slot_c ^= 1;
    };
    const std::true_type Y{};
    const std::false_type N{};
    step(0, Y, N, N);
    if (NG > 1) step(1, Y, Y, N);
    for (int it = 2; it < NG; ++it) step(it, Y, Y, Y);
    if (NG > 1) step(NG, N, Y, Y);
    else step(1, N, Y, N);
    step(NG + 1, N, N, Y);
    if (p.fault && !(vmax < 65504.f)) p.fault[0] = 1;   // range guard of the hidden layer's split
    if constexpr (LN) range_raise(p.fault, xmax);       // ... and of the normalised input's

    // ---------------- out = res + oacc 2^-11 + b2: register e of lane half lh is token (e & 3) + 8 (e >> 2) + 4 lh, column = li
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const long long row0 = tok0 + (wave * TT + t) * 32;
        if (row0 >= p.T) return;
        const long long rows_left = p.T - row0;
        const int rows = rows_left < 32 ? (int)rows_left : 32;
        const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + row0 * C), 0, rows * C * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res + row0 * C), 0, rows * C * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float bias = p.b2[j * 32 + li];
            float r[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                r[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, (((e & 3) + 8 * (e >> 2) + 4 * lh) * C + j * 32 + li) * 4, 0, 0));
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = fmaf(oacc[t][j][e], 1.0f / 2048.0f, bias);
                v += r[e];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rs, (((e & 3) + 8 * (e >> 2) + 4 * lh) * C + j * 32 + li) * 4, 0, 0);
            }
        }
    }
}

// ---- LayerNorm + ONE linear (Swin: norm1 -> to_qkv, swin_transformer.py:66-70,195): out[T][n] = LN(x) . W^T (+ bias), fp32 out.
// The same register-resident tokens, here as the A operand (the two fragment layouts of the 32x32x16 MFMA are the same: lane =
// row or column, lane half = k half): D[token][column], so a wave's stores are 128 contiguous bytes per token row.  Weights: the
// W1 tile images above, GPS groups of 32 output columns per step through two LDS slots.
struct LnLinearParams {
    const float* X32;    // [T][C]
    const float* ln_g;
    const float* ln_b;
    long long T;
    const f16* Wt;       // two_linear_w1_tiles_kernel of W [n][C]
    const float* bias;   // [n] or null
    float* out;          // [T][ldc]
    int n, ldc;
    int* fault;
};

template <int C, int NW, int GPS>
// (C = 192 is built for ONE block per CU: held to 256 registers - two blocks - it spilled 172 bytes per lane and re-read token fragments from
// scratch inside its K loop; reordering the LayerNorm phase, a second read of the row, packing the f16 pairs early and scheduling barriers in
// both phases all left 168-228 bytes (round 6).  The kernel streams x and qkv at HBM rate either way.)
__global__ __launch_bounds__(NW * 64, C > 96 ? 1 : 2) void ln_linear_f16x3_kernel(const LnLinearParams p) {
    constexpr int NT = 3 * C / 32, KS1 = 3 * C / 16, KR = 2 * C / 16;
    constexpr int HB = NT * 2048, SB = GPS * HB, PI = SB / 1024, IPW = (PI + NW - 1) / NW;
    static_assert(2 * SB + 8 * C <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char lds[2 * SB];
    __shared__ __attribute__((aligned(16))) float lns[2 * C];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const long long row0 = ((long long)blockIdx.x * NW + wave) * 32;
    const int nsteps = p.n / (32 * GPS);
    for (int i = tid; i < C; i += NW * 64) {
        lns[i] = p.ln_g[i];
        lns[C + i] = p.ln_b[i];
    }
    __syncthreads();
    half8 xf[KR];
    unsigned xmax;
    {
        long long tok = row0 + li;
        if (tok >= p.T) tok = p.T - 1;
        xmax = token_fragments_ln<C>(p.X32 + tok * C, lns, lh, xf, 0u);
    }
    auto issue_step = [&](int it, int slot) {
        const char* src = (const char*)p.Wt + (long long)it * SB;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            int q = wave + NW * j;
            if (q >= PI) q -= PI;
            __builtin_amdgcn_global_load_lds(GPTR(src + q * 1024 + lane * 16), LPTR(lds + slot * SB + q * 1024), 16, 0, 0);
        }
    };
    issue_step(0, 0);
    if (nsteps > 1) issue_step(1, 1);
    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = li * 64 + (((kk * 2 + lh) ^ (li >> 2)) & 3) * 16;
    const long long rows_left = p.T - row0;
    const int rows = rows_left <= 0 ? 0 : (rows_left < 32 ? (int)rows_left : 32);
    const __amdgpu_buffer_rsrc_t o_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (rows ? row0 : 0) * p.ldc), 0, rows ? ((rows - 1) * p.ldc + p.n) * 4 : 0, 0x00020000);
    int slot = 0;
    for (int it = 0; it < nsteps; ++it) {
        // own pieces of step it landed (and this wave's stores of step it - 1 retired: they share the counter) -> barrier -> step
        // it + 1 into the other slot, which every wave has finished reading
        if (it == 0 && nsteps > 1) WAIT_VMCNT(IPW);
        else WAIT_VMCNT(0);
        RAW_BARRIER();
        if (it >= 1 && it + 1 < nsteps) issue_step(it + 1, slot ^ 1);
        const char* Ws = lds + slot * SB;
        f32x16 acc[GPS];
#pragma unroll
        for (int g = 0; g < GPS; ++g)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int g = 0; g < GPS; ++g) {
                const half8 wf = *(const half8*)(Ws + g * HB + (ks >> 1) * 2048 + foff[ks & 1]);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[ks % KR], wf, acc[g], 0, 0, 0);
            }
#pragma unroll
        for (int g = 0; g < GPS; ++g) {
            const int col = (it * GPS + g) * 32 + li;
            const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = fmaf(acc[g][e], 1.0f / 2048.0f, bias);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rs, (((e & 3) + 8 * (e >> 2) + 4 * lh) * p.ldc + col) * 4, 0, 0);
            }
        }
        slot ^= 1;
    }
    range_raise(p.fault, xmax);
}

template <int C, int NW, int AHEAD, int TT>
void launch_variant(reid_ctx* ctx, const TwoLinearParams& p, int act) {
    const unsigned grid = (unsigned)((p.T + NW * 32 * TT - 1) / (NW * 32 * TT));
    // the builds in use: MLP = LayerNorm + GELU, to_out -> post_proj = packed input, no activation (+ the two mixed forms for tests)
    if (p.X32 && act) hipLaunchKernelGGL((two_linear_f16x3_kernel<C, NW, true, AHEAD, TT, true>), dim3(grid), dim3(NW * 64), 0, ctx->stream, p);
    else if (p.X32) hipLaunchKernelGGL((two_linear_f16x3_kernel<C, NW, false, AHEAD, TT, true>), dim3(grid), dim3(NW * 64), 0, ctx->stream, p);
    else if (act) hipLaunchKernelGGL((two_linear_f16x3_kernel<C, NW, true, AHEAD, TT, false>), dim3(grid), dim3(NW * 64), 0, ctx->stream, p);
    else hipLaunchKernelGGL((two_linear_f16x3_kernel<C, NW, false, AHEAD, TT, false>), dim3(grid), dim3(NW * 64), 0, ctx->stream, p);
}

}  // namespace

bool two_linear_supported(const reid_ctx* ctx, long long T, int C, int hid) {
    return ctx->precision == 2 && ctx->swin_two_linear && (C == 96 || C == 192) && hid % 32 == 0 && hid <= 4 * C && T >= 1;
}

// input: a16 [T][2C] ([xh | xl']) - or, a16 null, LayerNorm(x32 [T][C]; ln_g, ln_b) made in the kernel; w1 [hid][C], w2 [C][hid] fp32
// (the blob's); res / out fp32 [T][C]
int launch_two_linear(reid_ctx* ctx, const _Float16* a16, long long T, int C, int hid, const float* w1, const float* b1, const float* w2,
                      const float* b2, int act, const float* res, float* out, const float* x32, const float* ln_g, const float* ln_b) {
    ARG_CHECK(two_linear_supported(ctx, T, C, hid) && b1 && b2 && res && out && (a16 || (x32 && ln_g && ln_b)));
    const int HG = 1;   // one 32-unit hidden tile per group (the tile images are written for any)
    const void* k1 = (const char*)w1 + 1;   // the tiled images live beside the plain split forms (keys: the blob address + 1)
    const void* k2 = (const char*)w2 + 1;
    auto tiles = [&](const void* key, const float* w, bool second, void** d) -> int {
        auto it = ctx->split_w.find(key);
        if (it == ctx->split_w.end()) {
            void* t;
            const long long total = (long long)hid * 3 * C;
            HIP_TRY(hipMalloc(&t, (size_t)total * 2));
            if (second) hipLaunchKernelGGL(two_linear_w2_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, w, hid, C, HG, (f16*)t);
            else hipLaunchKernelGGL(two_linear_w1_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, w, hid, C, HG, (f16*)t);
            LAUNCH_CHECK();
            it = ctx->split_w.emplace(key, t).first;
        }
        *d = it->second;
        return REID_OK;
    };
    void *t1, *t2;
    REID_TRY(tiles(k1, w1, false, &t1));
    REID_TRY(tiles(k2, w2, true, &t2));
    TwoLinearParams p;
    p.A = a16; p.X32 = a16 ? nullptr : x32; p.ln_g = ln_g; p.ln_b = ln_b; p.T = T; p.W1t = (const f16*)t1; p.W2t = (const f16*)t2; p.b1 = b1; p.b2 = b2; p.res = res; p.out = out; p.hid = hid;
    p.fault = ctx->fault;
    p.ablate = ctx->two_linear_ablate;
    prof_begin(ctx, REID_K_CONV_GEMM, 4.0 * T * C * hid, (double)T * C * 12.0 + 8.0 * C * hid);
    // blocks of four waves (128 tokens): at C = 96 two of them share a CU, so one's prologue / epilogue runs beside the other's steps
    if (C == 192) launch_variant<192, 4, 2, 1>(ctx, p, act);
    else launch_variant<96, 4, 2, 1>(ctx, p, act);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

bool ln_linear_supported(const reid_ctx* ctx, long long T, int C, int n) {
    return ctx->precision == 2 && ctx->swin_two_linear && (C == 96 || C == 192) && n % 96 == 0 && T >= 1;
}

// out [T][ldc] fp32 = LayerNorm(x32 [T][C]; ln_g, ln_b) . w^T (+ bias); w [n][C] fp32 (the blob's)
int launch_ln_linear(reid_ctx* ctx, const float* x32, const float* ln_g, const float* ln_b, long long T, int C, int n, const float* w,
                     const float* bias, float* out, int ldc) {
    ARG_CHECK(ln_linear_supported(ctx, T, C, n) && x32 && ln_g && ln_b && w && out && ldc >= n);
    const void* key = (const char*)w + 1;
    auto it = ctx->split_w.find(key);
    if (it == ctx->split_w.end()) {
        void* t;
        const long long total = (long long)n * 3 * C;
        HIP_TRY(hipMalloc(&t, (size_t)total * 2));
        hipLaunchKernelGGL(two_linear_w1_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, w, n, C, 1, (f16*)t);
        LAUNCH_CHECK();
        it = ctx->split_w.emplace(key, t).first;
    }
    LnLinearParams p;
    p.X32 = x32; p.ln_g = ln_g; p.ln_b = ln_b; p.T = T; p.Wt = (const f16*)it->second; p.bias = bias; p.out = out; p.n = n; p.ldc = ldc;
    p.fault = ctx->fault;
    prof_begin(ctx, REID_K_CONV_GEMM, 2.0 * T * C * n, (double)T * 4.0 * (C + n) + 6.0 * C * n);
    const unsigned grid = (unsigned)((T + 127) / 128);
    if (C == 96) {
        hipLaunchKernelGGL((ln_linear_f16x3_kernel<96, 4, 1>), dim3(grid), dim3(256), 0, ctx->stream, p);   // (three column groups per step: no faster)
    } else {
        hipLaunchKernelGGL((ln_linear_f16x3_kernel<192, 4, 1>), dim3(grid), dim3(256), 0, ctx->stream, p);
    }
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
