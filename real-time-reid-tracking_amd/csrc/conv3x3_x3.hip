// fp32-class 3x3 / stride-1 / pad-1 convolution on the f16 matrix pipe, second form (round 5): TWO independent blocks per CU.
//
// conv3x3_f16.hip's SPLIT build is one 12-wave block per CU (138 KB of LDS): eight MFMA waves and four loader waves in ONE barrier
// domain.  Stamps of that kernel at 1024 crops (tools/diag_conv_f16.py, layer 4): per (chunk, tap) tile a compute wave spends 1 350
// cycles in fragment reads + MFMAs (1 024 of matrix pipe per SIMD: after every barrier all eight waves compute addresses, read
// their first fragments and wait for them TOGETHER, ~300 cycles in which the pipe is idle) and 410-450 cycles at the barrier
// (waiting for the loaders): 1 760 cycles per 1 024 of pipe, and nothing else on the CU to fill the holes.
// Here a block is FOUR waves (256 x BN tile, a wave owns 64 rows x BN columns: 128 accumulator registers at BN = 128) with
// 32-channel chunks (halo 2 x 23 KB + four 8-KB weight slots = 78 KB), so two blocks share a CU and each one's barrier waits,
// fragment-read latencies, prologue and epilogue lie under the other's MFMAs.  No loader waves: a wave issues its share of the
// tile three steps ahead (2 weight pieces + at most one halo piece per tile) right after the barrier, counted vmcnt.
//
// Arithmetic: x.w = xh.wh + (xl'.wh + xh.wl') 2^-11 as ONE fp32 accumulation over the weight parts [wh 2^11 | wh | wl']
// (Gemm16Params), per real 32-channel chunk c in the order  xh_c.(wh 2^11), xh_c.wl'  (halo buffer 0, 18 tiles),  xl'_c.wh  (halo
// buffer 1, 9 tiles): the xh halo is loaded once for its two products (conv3x3_f16.hip walks the 3 C virtual channels and loads it
// twice).  Another summation order than that kernel's, the same three products and roundings.
// Reference shapes: reid/backbones/SERes18_IBN.py:120-128 (BasicBlock convs), :250-276.
#include "reid_internal.h"
#include "conv3x3_geom.h"
#include "lin_math.h"
#include <type_traits>
#include <utility>

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(uintptr_t)(p))
#define RAW_BARRIER() asm volatile("s_barrier" ::: "memory")
#define LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int N>
__device__ __forceinline__ void lgkm_wait2(half8& x, half8& y) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N)); }
template <int N>
__device__ __forceinline__ void lgkm_wait1(half8& x) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N)); }

__device__ __forceinline__ void wait_vm(int n) {   // s_waitcnt takes an immediate
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int TW, int IMGS, int BN>
__global__ __launch_bounds__(256, 2) void conv3x3_x3_kernel(const Gemm16Params p) {
    constexpr int TH = 256 / (IMGS * TW);            // tile rows per image
    constexpr int WP = TW + 2, HP = TH + 2;          // halo pitch / rows
    constexpr int NPX = IMGS * HP * WP;              // halo pixels per block
    constexpr int NPI = (NPX + 15) / 16;             // halo pieces: 16 pixels x 64 B (32 channels)
    static_assert(NPI <= 24, "one halo piece per wave and tile over six tiles");
    constexpr int HALO_BYTES = NPI * 1024;
    constexpr int B_BYTES = BN * 64;                 // weight tile [BN][32]
    constexpr int BJ = BN / 64;                      // weight pieces per wave per tile
    constexpr int TM = 2, TN = BN / 32;              // wave tile 64 rows x BN columns
    constexpr int NSLOT = 4;
    static_assert(2 * (2 * HALO_BYTES + NSLOT * B_BYTES) <= 160 * 1024, "two blocks per CU");
    __shared__ __attribute__((aligned(16))) char lds[2 * HALO_BYTES + NSLOT * B_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    const int nnt = p.N / BN;
    int mtile, ntile;
    {   // XCD-aware, bijective block remap (blocks b and b+8 share an XCD): the N tiles of one M tile (same halo) sit on one XCD
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        mtile = L / nnt;
        ntile = L - mtile * nnt;
    }
    const int n_blk = ntile * BN;
    const int tiles_per_img = p.H / TH;              // IMGS == 2 -> 1
    const int img0 = IMGS == 2 ? mtile * 2 : mtile / tiles_per_img;
    const int y0 = IMGS == 2 ? 0 : (mtile - img0 * tiles_per_img) * TH;
    const int n_img = p.M / (p.H * p.W);
    const int C = p.Cin / 3;                         // real channels; A = [xh | xl'] (2 C per pixel), weights [wh 2^11 | wh | wl'] per tap
    const int a_cin = 2 * C;
    const int ncr = C / 32;
    const int nt = ncr * 27;

    const unsigned halo32 = (unsigned)(uintptr_t)lds, ring32 = halo32 + 2 * HALO_BYTES;

    // ---- halo pieces: piece q covers halo pixels 16 q .. +16 (lane / 4), 16-byte position lane % 4 (source-side XOR swizzle)
    auto issue_halo_piece = [&](int q, int vchunk, int buf) {
        const int hp = q * 16 + (lane >> 2);
        const int im = hp / (HP * WP), rem = hp - im * (HP * WP);
        const int hy = rem / WP, hx = rem - hy * WP;
        const int gy = y0 - 1 + hy, gx = hx - 1, gi = img0 + im;
        const bool ok = hp < NPX && gi < n_img && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const int cg = (lane & 3) ^ ((hp >> 2) & 3);
        const f16* src = ok ? p.A + (((long long)gi * p.H + gy) * p.W + gx) * a_cin + vchunk * 32 + cg * 8 : p.zero_page;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(lds + buf * HALO_BYTES + q * 1024), 16, 0, 0);
    };
    // ---- weight pieces: piece (wm * BJ + j) = rows 16 .. of the tile, lane / 4 = row, lane % 4 = position
    long long b_base[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = (wm * BJ + j) * 16 + (lane >> 2);
        const int cg = (lane & 3) ^ ((row >> 2) & 3);
        b_base[j] = (long long)(n_blk + row) * p.ldb + cg * 8;
    }
    auto issue_w = [&](int c, int r, int slot) {     // tile (chunk c, step r): part 0 (wh 2^11) r < 9, part 2 (wl') r < 18, part 1 (wh) else
        const int part = r < 9 ? 0 : r < 18 ? 2 : 1;
        const int tap = r < 9 ? r : r < 18 ? r - 9 : r - 18;
        const int k0 = tap * p.Cin + part * C + c * 32;
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            __builtin_amdgcn_global_load_lds(GPTR(p.B + b_base[j] + k0), LPTR(lds + 2 * HALO_BYTES + slot * B_BYTES + (wm * BJ + j) * 1024), 16, 0, 0);
    };

    // ---- A fragments: halo pixel of this lane's row in MFMA tile a at tap (0,0); B fragments: this lane's column
    int hp0[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        int im, y, x;
        row_to_pixel<TW, IMGS>(wm, a, li, im, y, x);
        hp0[a] = im * HP * WP + y * WP + x;
    }
    unsigned bx[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) bx[kk] = ring32 + li * 64 + (((kk * 2 + lh) ^ ((li >> 2) & 3)) * 16);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // ---- prologue: xh of chunk 0 (every piece), weight tiles 0 .. 2
    for (int q = wm; q < 24; q += 4) issue_halo_piece(q < NPI ? q : NPI - 1, 0, 0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    issue_w(0, 2, 2);

    int c = 0, r = 0, c3 = 0, r3 = 3, h1 = 0, h2 = 0;
    for (int t = 0; t < nt; ++t) {
        // in-order landing: everything but the pieces of the last two steps (tiles t + 1, t + 2 and their halo pieces) has landed,
        // i.e. this wave's share of tile t and of every older halo piece
        wait_vm(t + 2 >= nt ? 0 : 2 * BJ + h1 + h2);
        RAW_BARRIER();              // ... and every wave's share; every wave has finished tile t - 1: its slot and halo buffer are free
        if (t + 3 < nt) issue_w(c3, r3, (t + 3) & 3);
        const int hh = (r < 6 || (r >= 18 && r < 24)) ? 1 : 0;
        if (hh) {
            // steps 0 .. 5: xl' of this chunk into buffer 1 (last read in step 26 of the previous chunk); steps 18 .. 23: xh of the next
            // chunk into buffer 0 (last read in step 17; after the last chunk the same chunk again, read by nobody: every wave issues
            // exactly one piece per such step, so that the counted waits stay uniform)
            const int q = (r < 6 ? r : r - 18) * 4 + wm;
            const int cn = c + 1 < ncr ? c + 1 : c;
            issue_halo_piece(q < NPI ? q : NPI - 1, r < 6 ? ncr + c : cn, r < 6 ? 1 : 0);
        }
        {   // ---- tile (c, r): 2 k-steps of 16, 2 A + TN B fragments each, every read issued up front, waits counted per MFMA
            const int tap = r < 9 ? r : r < 18 ? r - 9 : r - 18;
            const int ty = tap / 3, tx = tap - ty * 3;
            const unsigned abuf = halo32 + (r >= 18 ? HALO_BYTES : 0);
            const unsigned boff = (unsigned)((t & 3) * B_BYTES);
            half8 fa[2][TM], fb[2][TN];
            unsigned aa[TM][2];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int hp = hp0[a] + ty * WP + tx;
                const unsigned base = abuf + hp * 64;
                const int swz = (hp >> 2) & 3;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) aa[a][kk] = base + (((kk * 2 + lh) ^ swz) * 16);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                LDS_READ(fa[kk][0], aa[0][kk], 0);
                const unsigned ba = bx[kk] + boff;
                LDS_READ(fb[kk][0], ba, 0);
                LDS_READ(fb[kk][1], ba, 2048);
                if constexpr (TN == 4) {
                    LDS_READ(fb[kk][2], ba, 4096);
                    LDS_READ(fb[kk][3], ba, 6144);
                }
                LDS_READ(fa[kk][1], aa[1][kk], 0);
            }
            // Reads return in order: of a k-step's PER reads fragment A0 is read 0, B_b read 1 + b, A1 the last.  Every wait names the
            // fragments it releases as in/out operands, so the MFMAs that consume them cannot be scheduled in front of it.
            constexpr int PER = TM + TN;
#define MM0(kk, b)                                                                                             \
    do {                                                                                                       \
        lgkm_wait2<((kk) == 0 ? 2 * PER : PER) - 2 - (b)>(fa[kk][0], fb[kk][b]);                               \
        acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][0], fb[kk][b], acc[0][b], 0, 0, 0);          \
    } while (0)
#define MM1(kk)                                                                                                \
    do {                                                                                                       \
        lgkm_wait1<((kk) == 0 ? PER : 0)>(fa[kk][1]);                                                          \
        _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                         \
            acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][1], fb[kk][b], acc[1][b], 0, 0, 0);      \
    } while (0)
            MM0(0, 0); MM0(0, 1);
            if constexpr (TN == 4) { MM0(0, 2); MM0(0, 3); }
            MM1(0);
            MM0(1, 0); MM0(1, 1);
            if constexpr (TN == 4) { MM0(1, 2); MM0(1, 3); }
            MM1(1);
#undef MM0
#undef MM1
            __builtin_amdgcn_sched_barrier(0);
        }
        h2 = h1;
        h1 = hh;
        if (++r == 27) { r = 0; ++c; }
        if (++r3 == 27) { r3 = 0; ++c3; }
    }
    __syncthreads();   // every wave is done with the last tile: LDS is free for the statistics

    if (p.ablate & 32) return;    // experiment: no epilogue
    // ------------------------------------------------------------------ fp32 epilogue, as conv3x3_f16.hip's SPLIT build (same
    // arithmetic, operation for operation): BN scale (x 2^-11) and shift, fp32 residual, ReLU from column relu_from on, fp32 or
    // [yh | yl'] stores, per-128-row column sums.  A wave owns 64 rows x BN columns here.
    const int ldc = (int)p.ldc;
    const int m_blk = mtile * 256;
    const int m_valid = p.M - m_blk;
    float s1[TN], s2[TN];
    float vmax = 0.f;
    const bool lean = (long long)256 * ldc * 4 < 0x7fffff00ll && (long long)256 * 2 * p.N * 2 < 0x7fffff00ll && (m_valid >= 256 || m_valid == 128);
    if (lean) {
        const bool wave_live = m_valid >= 256 || wm < 2;
        const int lrow[2] = {c_row_lane<TW, IMGS>(lh, 0), c_row_lane<TW, IMGS>(lh, 1)};
        const int col0 = n_blk + li;
        const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C32 + (long long)m_blk * ldc), 0, 256 * ldc * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)((p.res32 ? p.res32 : p.C32) + (long long)m_blk * ldc), 0, 256 * ldc * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t k_rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)((p.pack16 ? p.pack16 : (f16*)p.C32) + (long long)m_blk * 2 * p.N), 0, 256 * 2 * p.N * 2, 0x00020000);
        int voff[2], koff[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            voff[x] = (lrow[x] * ldc + col0) * 4;
            koff[x] = (lrow[x] * 2 * p.N + col0) * 2;
        }
        auto run = [&](auto res_c) {
            constexpr bool RES = decltype(res_c)::value;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = col0 + b * 32;
                const float cs = (p.col_scale ? p.col_scale[col] : 1.f) * p.acc_scale;
                const float sh = p.col_scale ? p.col_shift[col] : 0.f;
                const float lo = (p.relu && col >= p.relu_from) ? 0.f : -INFINITY;
                const bool pk = p.pack16 && col >= p.pack_from;      // uniform per 32-column block
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    float rr[16];
                    if constexpr (RES) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int x = ((e >> 2) == 1 || (e >> 2) == 2) ? 1 : 0;
                            const int urow = c_row_uniform<TW, IMGS>(wm, a, e);
                            rr[e] = wave_live ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, voff[x] + b * 128, urow * ldc * 4, 0)) : 0.f;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int x = ((e >> 2) == 1 || (e >> 2) == 2) ? 1 : 0;
                        const int urow = c_row_uniform<TW, IMGS>(wm, a, e);
                        float v = acc[a][b][e] * cs + sh;
                        v += RES ? rr[e] : 0.f;
                        v = fmaxf(v, lo);
                        if (wave_live) {
                            t1 += v;
                            t2 += v * v;
                            if (pk) {
                                vmax = fmaxf(vmax, fabsf(v));
                                const f16 hv = (f16)v;
                                const f16 lv = (f16)((v - (float)hv) * 2048.0f);
                                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), k_rs, koff[x] + b * 64, urow * 2 * p.N * 2, 0);
                                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, lv), k_rs, koff[x] + b * 64 + p.N * 2, urow * 2 * p.N * 2, 0);
                            } else if (!(p.ablate & 64)) {
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rs, voff[x] + b * 128, urow * ldc * 4, 0);
                            }
                        }
                    }
                }
                s1[b] = t1;
                s2[b] = t2;
            }
        };
        if (p.res32) run(std::true_type{});
        else run(std::false_type{});
    } else {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n_blk + b * 32 + li;
            const float cs = (p.col_scale ? p.col_scale[col] : 1.f) * p.acc_scale;
            const float sh = p.col_scale ? p.col_shift[col] : 0.f;
            const float lo = (p.relu && col >= p.relu_from) ? 0.f : -INFINITY;
            const bool pk = p.pack16 && col >= p.pack_from;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = c_row_natural<TW, IMGS>(wm, a, e, lh);
                    const int rc = row < m_valid ? row : 0;
                    float v = acc[a][b][e] * cs + sh;
                    v += p.res32 ? p.res32[(long long)(m_blk + rc) * ldc + col] : 0.f;
                    v = fmaxf(v, lo);
                    if (row < m_valid) {
                        t1 += v;
                        t2 += v * v;
                        if (pk) {
                            vmax = fmaxf(vmax, fabsf(v));
                            const f16 hv = (f16)v;
                            f16* dst = p.pack16 + (long long)(m_blk + row) * 2 * p.N + col;
                            dst[0] = hv;
                            dst[p.N] = (f16)((v - (float)hv) * 2048.0f);
                        } else {
                            p.C32[(long long)(m_blk + row) * ldc + col] = v;
                        }
                    }
                }
            }
            s1[b] = t1;
            s2[b] = t2;
        }
    }
    if (p.fault && !(vmax < 65504.f)) p.fault[0] = 1;   // a packed activation f16 cannot hold: the context reports it
    if (p.stats) {   // per 128 natural rows: waves 0,1 own rows 0..127, waves 2,3 rows 128..255 in every geometry
        float* stat_lds = (float*)lds;  // [4][BN][2]
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int lcol = b * 32 + li;
            const float t1 = s1[b] + __shfl_xor(s1[b], 32);
            const float t2 = s2[b] + __shfl_xor(s2[b], 32);
            if (lh == 0) {
                stat_lds[(wm * BN + lcol) * 2 + 0] = t1;
                stat_lds[(wm * BN + lcol) * 2 + 1] = t2;
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * BN; t += 256) {
            const int half = t / BN, cc = t - half * BN;
            if (half * 128 >= m_valid) continue;
            float* o = p.stats + ((long long)(mtile * 2 + half) * p.N + n_blk + cc) * 2;
            o[0] = stat_lds[((half * 2) * BN + cc) * 2 + 0] + stat_lds[((half * 2 + 1) * BN + cc) * 2 + 0];
            o[1] = stat_lds[((half * 2) * BN + cc) * 2 + 1] + stat_lds[((half * 2 + 1) * BN + cc) * 2 + 1];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same block structure on v_mfma_f32_16x16x32_f16.  Why: on RANDOM operands the chip does not hold its clock under
// v_mfma_f32_32x32x16_f16 - a registers-only loop of it runs at 1.55-1.6 PF, the 16x16x32 form at 1.95-2.0 PF (a quarter of the
// accumulator traffic per multiply), with the convolution's fragment reads and DMA pieces beside them 1.33 against 1.57 PF
// (tools/probes/overlap.hip, profiles/r05_overlap_probe.txt).  One 32-channel chunk is exactly one MFMA deep.
//
// Fragment maps.  A / B operand: lane l holds row (column) l & 15, channels 8 (l >> 4) .. + 8: ONE ds_read_b128 per lane and 16-row
// tile.  ds_read_b128 serves a wave in four fixed groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32):
// a group holds all 16 rows, the rows 4 .. 11 with channel group g0 ^ 1 and the others with g0.  An LDS row is 64 bytes (32
// channels), so a 256-byte bank window is four rows x four 16-byte positions; conflict-free means: rows with equal (index mod 4)
// sit at different positions.  Image: position of channel group g in row idx = g ^ 2 ((idx >> 3) & 1); MFMA row i <-> the tile's
// pixel PI(i) = i with the third and fourth quad swapped (i < 8: i, 8 .. 11: i + 4, 12 .. 15: i - 4), so that "g0 ^ 1" falls on the
// pixels whose (offset >> 2) is odd.  Then the four rows of a residue class (offsets o, o + 4, o + 8, o + 12 - or, in the 8-wide
// geometry, the rows y and y + 4 of the image: halo offsets 0 .. 7 and 40 .. 47, 40 = 8 mod 16) read positions
// g0 ^ ((u0 + k) & 2) ^ (k & 1), k = 0 .. 3: four different values for every u0, i.e. at every tap.
// C / D: lane l holds column l & 15, rows 4 (l >> 4) + reg: four rows x 64-byte segments per store instruction.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int pi16(int i) { return i < 8 ? i : (i < 12 ? i + 4 : i - 4); }

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// sum over the 16 lanes of a DPP row (every lane of the row gets it): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
}

template <int N>
__device__ __forceinline__ void lgkm_wait0() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// Everything behind the main loop of the 16x16x32 kernels: split-K reduction (last arriver), fp32 epilogue, statistics.
// COH (chain kernels, below): 16 = every load / store of an activation tensor carries sc1 - written through to / read from the point all
// XCDs share - because producer and consumer are blocks of ONE launch on different XCDs (no kernel boundary flushes the L2s between them)
template <int TW, int IMGS, int BN, bool LIN = false, int COH = 0>
__device__ __forceinline__ void x3m16_tail(const Gemm16Params& p, f32x4 (&acc)[4][BN / 16], char* lds, int tid, int wm, int mtile, int n_blk,
                                           int tile_id, int ksplit, int SK) {
    constexpr int TM = 4, TN = BN / 16;
    const int lane = tid & 63;
    const int l16 = lane & 15, lq = lane >> 4;
    const int pj = pi16(l16);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results (inline asm: hipcc does not count their wait states)
    // ... and nothing tied those wait states to the ACCUMULATORS: to the compiler acc[][] is ready the moment the last MFMA statement ends, so
    // a read of it may be scheduled in front of the s_nops (a "memory" clobber orders memory, not registers).  asm volatile statements keep
    // their order among themselves, so an empty one that names every accumulator as in / out pins all later reads behind the wait states.
    // (Round 5 saw "wrong sums" when the epilogue body was a nested generic lambda and reshaped the code until they went away - the
    // likeliest cause is exactly such a hoisted read; round 6 met the mirror case, a v_pk_mul feeding an inline-asm MFMA without wait
    // states in lin_x3_kernel: stale operands.  Both directions are now explicit.)
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        if constexpr (TN == 8)
            asm volatile("" : "+v"(acc[a][0]), "+v"(acc[a][1]), "+v"(acc[a][2]), "+v"(acc[a][3]), "+v"(acc[a][4]), "+v"(acc[a][5]), "+v"(acc[a][6]), "+v"(acc[a][7]));
        else
            asm volatile("" : "+v"(acc[a][0]), "+v"(acc[a][1]), "+v"(acc[a][2]), "+v"(acc[a][3]));
    }
    __syncthreads();

    // Split-K (few output tiles: a tracking frame, a pass of a few camera frames) as a REDUCE-SCATTER.  Round 5's form - every block
    // stores its partial tile, the block that arrives last re-reads all SK of them with dword loads and runs the whole epilogue - made
    // these kernels 2-3.6x SLOWER than conv3x3_f16.hip's on a 30-crop frame: with 128 accumulator registers live the last arriver's
    // 128 x SK loads went out in small batches, ~60 memory round trips on ONE CU per tile while the tile's other SK - 1 CUs had left.
    // Now: every block stores its partial as 16-byte units [a][b][lane] (device scope, written through), the SK blocks of a tile meet
    // at an arrival counter (below), and block k reduces and finishes column slice k of the tile - TN / SK sixteen-column units, the SK
    // partials of a unit requested together and summed in split order (deterministic) - epilogue, stores and column sums included.
    // Per block 128 / SK KB x SK of loads instead of 128 x SK KB on one; nothing is re-read.
    bool sk_path = false;
    if (SK > 1) {
        sk_path = true;
        constexpr int PART_B = 256 * BN * 4;       // bytes of one partial tile
        char* part = (char*)(p.splitk_ws + (long long)tile_id * SK * (256 * BN));
        const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)part, 0, SK * PART_B, 0x00020000);
        // ... except the units of this block's OWN column slice: nobody else reads them, so they wait in LDS (the main loop's buffers are free;
        // behind the 4 KB of the column sums) instead of making the round trip through memory - a quarter of the partial traffic at SK = 4,
        // half of it at SK = 2.  A lane reads back exactly what it wrote: no barrier.
        const int nb_own = TN / SK;
        const unsigned own32 = (unsigned)(uintptr_t)lds + 4096u + (unsigned)tid * 16u;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                if (b / nb_own == ksplit)
                    asm volatile("ds_write_b128 %0, %1" ::"v"(own32 + (unsigned)((a * nb_own + (b - ksplit * nb_own)) * 4096)), "v"(acc[a][b]) : "memory");
                else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[a][b]), w_rs, tid * 16, ksplit * PART_B + (a * TN + b) * 4096, 16 /* sc1 */);
            }
        // sc1 stores are written through to the point all XCDs share, sc1 loads bypass this XCD's L2 (conv3x3_f16.hip: no L2 write-back /
        // invalidate fence); acknowledged once vmcnt reaches 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            // The tile's SK blocks have consecutive linear ids on one XCD (the remap above): blocks are dispatched in id order, so a block's
            // mates are resident or next in line, and a launch of <= 512 of these blocks is resident as a whole.  The wait is bounded
            // all the same: a rendezvous that never completes raises the context's fault word instead of hanging the device.
            __hip_atomic_fetch_add(p.splitk_cnt + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(p.splitk_cnt + tile_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < SK) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 23)) {
                    if (p.fault) p.fault[2] = 1;
                    break;
                }
            }
        }
        __syncthreads();
    }

    if (p.ablate & 32) return;
    // ------------------------------------------------------------------ fp32 epilogue: the arithmetic of conv3x3_f16.hip's SPLIT
    // build per element (BN scale x 2^-11 + shift, + fp32 residual, ReLU from relu_from on, fp32 or [yh | yl'] stores, per-128-row
    // column sums).  The accumulators are TRANSPOSED tiles (the MFMAs take the weight fragment as their first operand): lane
    // (l16, lq) holds the FOUR CONSECUTIVE CHANNELS 16 b + 4 PQ(lq) + e of pixel PI(l16) of 16-pixel tile a (PQ = quads 2 and 3
    // swapped) - so a lane's four values are 16 contiguous bytes of the NHWC output: one buffer_store_dwordx4 (or two dwordx2 of
    // the [yh | yl'] form) and one dwordx4 residual load per tile instead of four (eight) scalar ones.  Column sums: per-lane
    // partials over the four tiles, then four DPP adds across the 16 pixel lanes of a row.
    const int ldc = (int)p.ldc;
    const int m_blk = mtile * 256;
    const int m_valid = p.M - m_blk;
    const bool wave_live = wm * 64 < m_valid;               // convolutions: M % 128 == 0 (a ragged tile is the rows of waves 0, 1); linears: any M
    const int pq = lq < 2 ? lq : 5 - lq;
    const int lrow = TW == 8 ? 32 * (pj >> 3) + (pj & 7) : pj;                       // lane part of the natural row (tile a: ubase)
    const int col0 = n_blk + 4 * pq;
    const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C32 + (long long)m_blk * ldc), 0, 256 * ldc * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)((p.res32 ? p.res32 : p.C32) + (long long)m_blk * ldc), 0, 256 * ldc * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t k_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)((p.pack16 ? p.pack16 : (f16*)p.C32) + (long long)m_blk * 2 * p.N), 0, 256 * 2 * p.N * 2, 0x00020000);
    // rows past M (ragged linear tiles): the row term of a store / residual address is in the SCALAR offset, which the descriptor's
    // range check does not see - so such a row gets a lane offset past the descriptor instead (stores dropped, loads zero)
    int voff_a[TM], koff_a[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const bool row_ok = wm * 64 + (TW == 8 ? a * 8 : a * 16) + lrow < m_valid;
        voff_a[a] = row_ok ? (lrow * ldc + col0) * 4 : 0x7fffff00;
        koff_a[a] = row_ok ? (lrow * 2 * p.N + col0) * 2 : 0x7fffff00;
    }
    const bool has_stats = p.stats != nullptr;
    float vmax = 0.f;          // largest packed magnitude of the unsplit path (after the ReLU, as in conv3x3_f16.hip: a NaN that a ReLU column
                               // clamps to 0 is caught downstream - the neck's non-finite check - not here; the bit-pattern form taken before the
                               // clamp, which the split-K path below uses, costs this hot epilogue 1 % of a 1024-crop pass)
    // every choice below (residual, [yh | yl'] or fp32 store, statistics) is uniform per wave and per 16-column tile and is taken
    // by a scalar branch AROUND a straight-line body of four tiles: per-element selects and branches made this epilogue ~1 000
    // vector instructions per wave, each of which waits 11-45 cycles for an issue slot beside the other blocks' MFMAs
    // (profiles/r02_coissue.log; s_memtime stamps, round 5: 28 k cycles of a 64-wide block's 74 k-cycle life).  One flat lambda on
    // purpose: the same body as a nested generic lambda per tile compiled (ROCm 7.2 hipcc) to code that gave wrong sums.
    auto run = [&](auto res_c, auto st_c) {
        constexpr bool RES = decltype(res_c)::value, ST = decltype(st_c)::value;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = col0 + b * 16, tcol = n_blk + b * 16;
            f32x4 cs = f32x4{p.acc_scale, p.acc_scale, p.acc_scale, p.acc_scale}, sh = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.col_scale) {
                cs = *(const f32x4*)(p.col_scale + col) * p.acc_scale;
                sh = *(const f32x4*)(p.col_shift + col);
            } else if (LIN && p.col_shift) {      // linear layers: bias only
                sh = *(const f32x4*)(p.col_shift + col);
            }
            const bool gelu = LIN && p.act == 1;  // linear layers: out = act(acc + bias) + residual (gemm_f16.hip's linear epilogue, lin_math.h)
            const float lo = (p.relu && tcol >= p.relu_from) ? 0.f : -INFINITY;   // uniform per 16-column tile (relu_from % 16 == 0)
            const bool pk = p.pack16 && tcol >= p.pack_from;                       // (pack_from % 32 == 0)
            // AB tiles at a time: all four for the 64-wide block (its 64 accumulator registers leave room; the four residual
            // loads are in flight together), one for the 128-wide block (128 accumulators: more would spill)
            constexpr int AB = BN == 64 ? 4 : 1;
            f32x4 t1 = f32x4{0.f, 0.f, 0.f, 0.f}, t2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a0 = 0; a0 < TM; a0 += AB) {
                f32x4 v[AB];
#pragma unroll
                for (int i = 0; i < AB; ++i) {
                    const int a = a0 + i;
                    v[i] = acc[a][b] * cs + sh;
                    if (gelu) {      // two values per instruction: bit-identical to the scalar function (lin_math.h)
                        const gelu_f32x2 g0 = gelu2_f16_storage(gelu_f32x2{v[i][0], v[i][1]}), g1 = gelu2_f16_storage(gelu_f32x2{v[i][2], v[i][3]});
                        v[i] = f32x4{g0.x, g0.y, g1.x, g1.y};
                    }
                    if constexpr (RES)
                        v[i] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, voff_a[a] + b * 64, (wm * 64 + (TW == 8 ? a * 8 : a * 16)) * ldc * 4, COH));
                    if (!LIN) {      // (linear layers have no ReLU)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], lo);
                    }
                    if constexpr (ST) {
                        t1 += v[i];
                        t2 += v[i] * v[i];
                    }
                }
                if (pk) {
#pragma unroll
                    for (int i = 0; i < AB; ++i) {
                        const int ubase = wm * 64 + (TW == 8 ? (a0 + i) * 8 : (a0 + i) * 16);       // uniform part of the natural row
                        u32x2 hw, lw;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float v0 = v[i][2 * h], v1 = v[i][2 * h + 1];
                            vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));
                            const f16 h0 = cvt_f16_rn(v0), h1 = cvt_f16_rn(v1);      // one rounding of the materialised fp32 value (lin_math.h)
                            const f16 l0 = cvt_f16_rn((v0 - (float)h0) * 2048.0f), l1 = cvt_f16_rn((v1 - (float)h1) * 2048.0f);
                            hw[h] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                            lw[h] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                        }
                        __builtin_amdgcn_raw_buffer_store_b64(hw, k_rs, koff_a[a0 + i] + b * 32, ubase * 2 * p.N * 2, COH);
                        __builtin_amdgcn_raw_buffer_store_b64(lw, k_rs, koff_a[a0 + i] + b * 32 + p.N * 2, ubase * 2 * p.N * 2, COH);
                    }
                } else if (!(p.ablate & 64)) {
#pragma unroll
                    for (int i = 0; i < AB; ++i)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i]), c_rs, voff_a[a0 + i] + b * 64,
                                                               (wm * 64 + (TW == 8 ? (a0 + i) * 8 : (a0 + i) * 16)) * ldc * 4, COH);
                }
            }
            if constexpr (ST) {   // this wave's column sums of the tile's 64 rows: four DPP adds across the 16 pixel lanes, lane 0 of the row writes
                float* stat_lds = (float*)lds;  // [4][BN][2]  (every wave is past the main loop's LDS reads and the split-K flag)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r1 = row16_sum(t1[e]), r2 = row16_sum(t2[e]);
                    if (l16 == 0) {
                        const int lcol = b * 16 + 4 * pq + e;
                        stat_lds[(wm * BN + lcol) * 2 + 0] = r1;
                        stat_lds[(wm * BN + lcol) * 2 + 1] = r2;
                    }
                }
            }
        }
    };
    // ---- split-K: this block's column slice (units b = ksplit TN / SK ..): the SK partials of a unit are requested together, summed
    // in split order, and go through the arithmetic of run() above, element for element (b is a run-time value here: the unit's
    // column offset sits in the address registers, the body is compiled once per (SK, residual, statistics))
    unsigned vm_sk = 0u;
    auto run_sk = [&](auto sk_c, auto res_c, auto st_c) {
        constexpr int SKC = decltype(sk_c)::value;
        constexpr bool RES = decltype(res_c)::value, ST = decltype(st_c)::value;
        constexpr int PART_B = 256 * BN * 4;
        const __amdgpu_buffer_rsrc_t w_rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.splitk_ws + (long long)tile_id * SKC * (256 * BN)), 0, SKC * PART_B, 0x00020000);
        constexpr int NB = TN / SKC;
        for (int bi = 0; bi < NB; ++bi) {
            const int b = ksplit * NB + bi;
            f32x4 v[TM];
            {
                u32x4 ld[SKC][TM];
                const unsigned own32 = (unsigned)(uintptr_t)lds + 4096u + (unsigned)tid * 16u;
#pragma unroll
                for (int sidx = 0; sidx < SKC; ++sidx)
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        if (sidx == ksplit) {     // this block's own partial of the unit: parked in LDS above
                            f32x4 t;
                            asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(own32 + (unsigned)((a * NB + bi) * 4096)) : "memory");
                            ld[sidx][a] = __builtin_bit_cast(u32x4, t);
                        } else {
                            ld[sidx][a] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, tid * 16, sidx * PART_B + (a * TN + b) * 4096, 16 /* sc1 */);
                        }
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the inline ds_read is not counted by the compiler)
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    v[a] = __builtin_bit_cast(f32x4, ld[0][a]);
#pragma unroll
                    for (int sidx = 1; sidx < SKC; ++sidx) v[a] += __builtin_bit_cast(f32x4, ld[sidx][a]);
                }
            }
            const int col = col0 + b * 16, tcol = n_blk + b * 16;
            f32x4 cs = f32x4{p.acc_scale, p.acc_scale, p.acc_scale, p.acc_scale}, sh = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.col_scale) {
                cs = *(const f32x4*)(p.col_scale + col) * p.acc_scale;
                sh = *(const f32x4*)(p.col_shift + col);
            }
            const float lo = (p.relu && tcol >= p.relu_from) ? 0.f : -INFINITY;
            const bool pk = p.pack16 && tcol >= p.pack_from;
            f32x4 t1 = f32x4{0.f, 0.f, 0.f, 0.f}, t2 = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 rr[TM];
            if constexpr (RES) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    rr[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, voff_a[a] + b * 64, (wm * 64 + (TW == 8 ? a * 8 : a * 16)) * ldc * 4, COH));
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                f32x4 x = v[a] * cs + sh;
                if constexpr (RES) x += rr[a];
                if (pk) vm_sk = range_acc(range_acc(range_acc(range_acc(vm_sk, x[0]), x[1]), x[2]), x[3]);   // before the ReLU: max(NaN, 0) is 0
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], lo);
                if constexpr (ST) {
                    t1 += x;
                    t2 += x * x;
                }
                const int ubase = wm * 64 + (TW == 8 ? a * 8 : a * 16);
                if (pk) {
                    u32x2 hw, lw;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float v0 = x[2 * h], v1 = x[2 * h + 1];
                        const f16 h0 = cvt_f16_rn(v0), h1 = cvt_f16_rn(v1);
                        const f16 l0 = cvt_f16_rn((v0 - (float)h0) * 2048.0f), l1 = cvt_f16_rn((v1 - (float)h1) * 2048.0f);
                        hw[h] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                        lw[h] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(hw, k_rs, koff_a[a] + b * 32, ubase * 2 * p.N * 2, COH);
                    __builtin_amdgcn_raw_buffer_store_b64(lw, k_rs, koff_a[a] + b * 32 + p.N * 2, ubase * 2 * p.N * 2, COH);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), c_rs, voff_a[a] + b * 64, ubase * ldc * 4, COH);
                }
            }
            if constexpr (ST) {
                float* stat_lds = (float*)lds;  // [4][BN][2]
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r1 = row16_sum(t1[e]), r2 = row16_sum(t2[e]);
                    if (l16 == 0) {
                        const int lcol = b * 16 + 4 * pq + e;
                        stat_lds[(wm * BN + lcol) * 2 + 0] = r1;
                        stat_lds[(wm * BN + lcol) * 2 + 1] = r2;
                    }
                }
            }
        }
    };
    auto run_sk_n = [&](auto sk_c) {
        if (p.res32) {
            if (has_stats) run_sk(sk_c, std::true_type{}, std::true_type{});
            else run_sk(sk_c, std::true_type{}, std::false_type{});
        } else {
            if (has_stats) run_sk(sk_c, std::false_type{}, std::true_type{});
            else run_sk(sk_c, std::false_type{}, std::false_type{});
        }
    };
    if (sk_path) {
        if constexpr (!LIN) {
            if (wave_live) {
                if (SK == 2) run_sk_n(std::integral_constant<int, 2>{});
                else if (SK == 4) run_sk_n(std::integral_constant<int, 4>{});
                else if constexpr (TN == 8) run_sk_n(std::integral_constant<int, 8>{});
            }
            range_raise(p.fault, vm_sk);
        }
    } else if (wave_live) {
        if (p.res32) {
            if (has_stats) run(std::true_type{}, std::true_type{});
            else run(std::true_type{}, std::false_type{});
        } else {
            if (has_stats) run(std::false_type{}, std::true_type{});
            else run(std::false_type{}, std::false_type{});
        }
    }
    if (p.fault && !(vmax < 65504.f)) p.fault[0] = 1;
    if (sk_path) {
        // every load of the partials has returned (their values were used above): the block that counts the tile's last reader
        // resets the two counters for the next launch
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(p.splitk_cnt + 512 + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == SK - 1) {
                __hip_atomic_store(p.splitk_cnt + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.splitk_cnt + 512 + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (has_stats) {   // per 128 natural rows: waves 0,1 own rows 0..127, waves 2,3 rows 128..255
        const float* stat_lds = (const float*)lds;  // [4][BN][2], written by run()
        __syncthreads();
        const int cc_lo = sk_path ? ksplit * (BN / SK) : 0, cc_hi = sk_path ? cc_lo + BN / SK : BN;   // split-K: this block's column slice
        for (int t = tid; t < 2 * BN; t += 256) {
            const int half = t / BN, cc = t - half * BN;
            if (half * 128 >= m_valid || cc < cc_lo || cc >= cc_hi) continue;
            float* o = p.stats + ((long long)(mtile * 2 + half) * p.N + n_blk + cc) * 2;
            const float o0 = stat_lds[((half * 2) * BN + cc) * 2 + 0] + stat_lds[((half * 2 + 1) * BN + cc) * 2 + 0];
            const float o1 = stat_lds[((half * 2) * BN + cc) * 2 + 1] + stat_lds[((half * 2 + 1) * BN + cc) * 2 + 1];
            if constexpr (COH != 0) {
                __hip_atomic_store(o, o0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 1, o1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                o[0] = o0;
                o[1] = o1;
            }
        }
    }
}

// One output tile (tile_id, split ksplit of SK) of the looped 16x16x32 kernel as a device function: the body of conv3x3_x3m16_kernel, and
// of the conv items of the chain kernels below (COH = 16: activations through sc1 accesses, the item's dependency wait between its
// first weight requests and its first halo request).
template <int TW, int IMGS, int BN, int COH, class WaitFn>
__device__ __forceinline__ void x3m16_item(const Gemm16Params& p, char* lds, int tile_id, int ksplit, int SK, WaitFn&& wait_deps) {
    constexpr int TH = 256 / (IMGS * TW);
    constexpr int WP = TW + 2, HP = TH + 2;
    constexpr int NPX = IMGS * HP * WP;
    constexpr int NPI = (NPX + 15) / 16;
    static_assert(NPI <= 24, "one halo piece per wave and tile over six tiles");
    constexpr int HALO_BYTES = NPI * 1024;
    constexpr int B_BYTES = BN * 64;
    constexpr int BJ = BN / 64;
    constexpr int TM = 4, TN = BN / 16;              // wave tile 64 rows x BN columns in 16 x 16 MFMA tiles
    // BN = 128 (layers 2-4): two blocks per CU, two halo buffers, four weight slots.  BN = 64 (layer 1, form 3): FOUR blocks per CU - a
    // layer-1 block is 54 tiles of 16 MFMAs per wave between a prologue and an epilogue that move 215 KB, so what it needs is other
    // blocks to run under its memory phases, not prefetch depth: ONE halo buffer (a phase change waits for its halo), 38.5 KB per block
    constexpr int NS = 4;
    constexpr int NHB = BN == 64 ? 1 : 2;            // halo buffers
    static_assert((BN == 64 ? 4 : 2) * (NHB * HALO_BYTES + NS * B_BYTES) <= 160 * 1024, "blocks per CU");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;       // row / column inside a 16-wide MFMA tile, channel group (operands) = row quad (C / D)

    const int nnt = p.N / BN;
    // Split-K (few output tiles - a tracking frame): SK blocks share one output tile, each summing C / SK of the real channels; the
    // one that arrives last adds the fp32 partials in split order (deterministic) and runs the epilogue (as conv3x3_f16.hip).
    const int mtile = tile_id / nnt, ntile = tile_id - mtile * nnt;
    const int n_blk = ntile * BN;
    const int tiles_per_img = p.H / TH;
    const int img0 = IMGS == 2 ? mtile * 2 : mtile / tiles_per_img;
    const int y0 = IMGS == 2 ? 0 : (mtile - img0 * tiles_per_img) * TH;
    const int n_img = p.M / (p.H * p.W);
    const int C = p.Cin / 3;
    const int a_cin = 2 * C;
    const int ncr_all = C / 32;
    const int ncr = ncr_all / SK, c0 = ksplit * ncr;     // real 32-channel chunks of this block: [c0, c0 + ncr)
    const int nt = ncr * 27;
    const unsigned halo32 = (unsigned)(uintptr_t)lds, ring32 = halo32 + NHB * HALO_BYTES;

    auto issue_halo_piece = [&](int q, int vchunk, int buf) {
        const int hp = q * 16 + (lane >> 2);
        const int im = hp / (HP * WP), rem = hp - im * (HP * WP);
        const int hy = rem / WP, hx = rem - hy * WP;
        const int gy = y0 - 1 + hy, gx = hx - 1, gi = img0 + im;
        const bool ok = hp < NPX && gi < n_img && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const int cg = (lane & 3) ^ (((hp >> 3) & 1) << 1);
        const f16* src = ok ? p.A + (((long long)gi * p.H + gy) * p.W + gx) * a_cin + vchunk * 32 + cg * 8 : p.zero_page;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(lds + buf * HALO_BYTES + q * 1024), 16, 0, COH);
    };
    long long b_base[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = (wm * BJ + j) * 16 + (lane >> 2);
        const int cg = (lane & 3) ^ (((row >> 3) & 1) << 1);
        b_base[j] = (long long)(n_blk + row) * p.ldb + cg * 8;
    }
    // ---- A: halo pixel of this lane's row in 16-row tile a at tap (0,0).  Tile a of wave wm = natural rows 64 wm + 16 a .. (32-
    // and 16-wide maps) or 64 wm + 8 a + {0 .. 7, 32 .. 39} (8-wide: image rows y and y + 4)
    const int pj = pi16(l16);                         // this lane's pixel inside the tile
    int hp0[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        if constexpr (TW == 32) hp0[a] = (2 * wm + (a >> 1)) * WP + 16 * (a & 1) + pj;
        else if constexpr (TW == 16) hp0[a] = (4 * wm + a) * WP + pj;
        else hp0[a] = (wm >> 1) * HP * WP + ((wm & 1) * 8 + a + 4 * (pj >> 3)) * WP + (pj & 7);
    }
    // ---- B: column b * 16 + PI(l16) of the weight tile
    const unsigned bx = ring32 + pj * 64 + ((lq ^ (((pj >> 3) & 1) << 1)) * 16);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue_w = [&](int c, int r, int slot) {
        const int part = r < 9 ? 0 : r < 18 ? 2 : 1;
        const int tap = r < 9 ? r : r < 18 ? r - 9 : r - 18;
        const int k0 = tap * p.Cin + part * C + (c0 + c) * 32;
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            __builtin_amdgcn_global_load_lds(GPTR(p.B + b_base[j] + k0), LPTR(lds + NHB * HALO_BYTES + slot * B_BYTES + (wm * BJ + j) * 1024), 16, 0, 0);
    };

    if constexpr (BN != 64) {
    if constexpr (COH != 0) {
        // chain kernels: the weights of the first three steps do not depend on the producer of this item's input - they are requested
        // before the item waits for its image(s); the halo follows, and step 0 waits for everything (the counted wait below assumes
        // the halo pieces were issued first)
        issue_w(0, 0, 0);
        issue_w(0, 1, 1);
        issue_w(0, 2, 2);
        wait_deps();
        for (int q = wm; q < 24; q += 4) issue_halo_piece(q < NPI ? q : NPI - 1, c0, 0);
    } else {
    for (int q = wm; q < 24; q += 4) issue_halo_piece(q < NPI ? q : NPI - 1, c0, 0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    issue_w(0, 2, 2);
    }

    int c = 0, r = 0, c3 = 0, r3 = 3, h1 = 0, h2 = 0;
    for (int t = 0; t < nt; ++t) {
        wait_vm((t + 2 >= nt || (COH != 0 && t == 0)) ? 0 : 2 * BJ + h1 + h2);
        RAW_BARRIER();
        if (t + 3 < nt) issue_w(c3, r3, (t + 3) & 3);
        const int hh = (r < 6 || (r >= 18 && r < 24)) ? 1 : 0;
        if (hh) {
            const int q = (r < 6 ? r : r - 18) * 4 + wm;
            const int cn = c + 1 < ncr ? c + 1 : c;
            issue_halo_piece(q < NPI ? q : NPI - 1, r < 6 ? ncr_all + c0 + c : c0 + cn, r < 6 ? 1 : 0);
        }
        {   // ---- tile (c, r): one MFMA deep; 4 A + TN B fragments, all requested up front, waits counted per MFMA row
            const int tap = r < 9 ? r : r < 18 ? r - 9 : r - 18;
            const int ty = tap / 3, tx = tap - ty * 3;
            const unsigned abuf = halo32 + (r >= 18 ? HALO_BYTES : 0);
            const unsigned ba = bx + (unsigned)((t & 3) * B_BYTES);
            half8 fa[TM], fb[TN];
            unsigned aa[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int hp = hp0[a] + ty * WP + tx;
                aa[a] = abuf + hp * 64 + ((lq ^ (((hp >> 3) & 1) << 1)) * 16);
            }
            // order: A0, B0 .. B(TN-1), A1, A2, A3
            LDS_READ(fa[0], aa[0], 0);
            LDS_READ(fb[0], ba, 0);
            LDS_READ(fb[1], ba, 1024);
            LDS_READ(fb[2], ba, 2048);
            LDS_READ(fb[3], ba, 3072);
            if constexpr (TN == 8) {
                LDS_READ(fb[4], ba, 4096);
                LDS_READ(fb[5], ba, 5120);
                LDS_READ(fb[6], ba, 6144);
                LDS_READ(fb[7], ba, 7168);
            }
            LDS_READ(fa[1], aa[1], 0);
            LDS_READ(fa[2], aa[2], 0);
            LDS_READ(fa[3], aa[3], 0);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NRD = TM + TN;
            // (inline asm: the waits are asm volatile statements, and only asm volatile statements keep their order among each other -
            // hipcc moved builtin MFMAs below later waits, through sched_barriers.  Accumulators are only ever MFMA SrcC / vDst inside
            // the loop: back-to-back issue needs no wait states; the epilogue reads them behind the loop's closing barrier + s_nops.)
#define MMA(a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[a][b]) : "v"(fb[b]), "v"(fa[a]))   /* weights first: D^T (x3m16_tail) */
#define MM0(b)                                   \
    do {                                         \
        lgkm_wait1<NRD - 2 - (b)>(fb[b]);        \
        MMA(0, b);                               \
    } while (0)
            lgkm_wait1<NRD - 1>(fa[0]);           // (every wait names what it releases: tying fa[0] to each of them made hipcc copy it per MFMA)
            MM0(0); MM0(1); MM0(2); MM0(3);
            if constexpr (TN == 8) { MM0(4); MM0(5); MM0(6); MM0(7); }
            lgkm_wait1<2>(fa[1]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(1, b);
            __builtin_amdgcn_sched_barrier(0);
            lgkm_wait1<1>(fa[2]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(2, b);
            __builtin_amdgcn_sched_barrier(0);
            lgkm_wait1<0>(fa[3]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(3, b);
#undef MM0
#undef MMA
            __builtin_amdgcn_sched_barrier(0);
        }
        h2 = h1;
        h1 = hh;
        if (++r == 27) { r = 0; ++c; }
        if (++r3 == 27) { r3 = 0; ++c3; }
    }
    } else {
        // ---- BN = 64: one halo buffer.  Per chunk: xh (18 tiles), then xl' (9 tiles); a phase starts with barrier (every wave has read
        // the old halo), this wave's six halo pieces, vmcnt(0); inside a phase only the weight ring is in flight (counted waits)
        issue_w(0, 0, 0);
        issue_w(0, 1, 1);
        issue_w(0, 2, 2);
        if constexpr (COH != 0) wait_deps();     // (chain kernels: the halo of the first phase is requested below, behind the wait)
        int c = 0, r = 0, c3 = 0, r3 = 3;
        for (int t = 0; t < nt; ++t) {
            const bool phase_start = r == 0 || r == 18;
            if (phase_start) {
                if (t > 0) RAW_BARRIER();
                const int vch = r == 0 ? c0 + c : ncr_all + c0 + c;
                for (int q = wm; q < 24; q += 4) issue_halo_piece(q < NPI ? q : NPI - 1, vch, 0);
            }
            wait_vm((phase_start || t + 2 >= nt) ? 0 : 2 * BJ);
            RAW_BARRIER();
            if (t + 3 < nt) issue_w(c3, r3, (t + 3) & 3);
        {   // ---- tile (c, r): one MFMA deep; 4 A + TN B fragments, all requested up front, waits counted per MFMA row
            const int tap = r < 9 ? r : r < 18 ? r - 9 : r - 18;
            const int ty = tap / 3, tx = tap - ty * 3;
            const unsigned abuf = halo32;
            const unsigned ba = bx + (unsigned)((t & 3) * B_BYTES);
            half8 fa[TM], fb[TN];
            unsigned aa[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int hp = hp0[a] + ty * WP + tx;
                aa[a] = abuf + hp * 64 + ((lq ^ (((hp >> 3) & 1) << 1)) * 16);
            }
            // order: A0, B0 .. B(TN-1), A1, A2, A3
            LDS_READ(fa[0], aa[0], 0);
            LDS_READ(fb[0], ba, 0);
            LDS_READ(fb[1], ba, 1024);
            LDS_READ(fb[2], ba, 2048);
            LDS_READ(fb[3], ba, 3072);
            if constexpr (TN == 8) {
                LDS_READ(fb[4], ba, 4096);
                LDS_READ(fb[5], ba, 5120);
                LDS_READ(fb[6], ba, 6144);
                LDS_READ(fb[7], ba, 7168);
            }
            LDS_READ(fa[1], aa[1], 0);
            LDS_READ(fa[2], aa[2], 0);
            LDS_READ(fa[3], aa[3], 0);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NRD = TM + TN;
            // (inline asm: the waits are asm volatile statements, and only asm volatile statements keep their order among each other -
            // hipcc moved builtin MFMAs below later waits, through sched_barriers.  Accumulators are only ever MFMA SrcC / vDst inside
            // the loop: back-to-back issue needs no wait states; the epilogue reads them behind the loop's closing barrier + s_nops.)
#define MMA(a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[a][b]) : "v"(fb[b]), "v"(fa[a]))   /* weights first: D^T (x3m16_tail) */
#define MM0(b)                                   \
    do {                                         \
        lgkm_wait1<NRD - 2 - (b)>(fb[b]);        \
        MMA(0, b);                               \
    } while (0)
            lgkm_wait1<NRD - 1>(fa[0]);           // (every wait names what it releases: tying fa[0] to each of them made hipcc copy it per MFMA)
            MM0(0); MM0(1); MM0(2); MM0(3);
            if constexpr (TN == 8) { MM0(4); MM0(5); MM0(6); MM0(7); }
            lgkm_wait1<2>(fa[1]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(1, b);
            __builtin_amdgcn_sched_barrier(0);
            lgkm_wait1<1>(fa[2]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(2, b);
            __builtin_amdgcn_sched_barrier(0);
            lgkm_wait1<0>(fa[3]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(3, b);
#undef MM0
#undef MMA
            __builtin_amdgcn_sched_barrier(0);
        }
            if (++r == 27) { r = 0; ++c; }
            if (++r3 == 27) { r3 = 0; ++c3; }
        }
    }
    x3m16_tail<TW, IMGS, BN, false, COH>(p, acc, lds, tid, wm, mtile, n_blk, tile_id, ksplit, SK);
}

template <int TW, int IMGS, int BN>
__global__ __launch_bounds__(256, BN == 64 ? 4 : 2) void conv3x3_x3m16_kernel(const Gemm16Params p) {
    constexpr int TH = 256 / (IMGS * TW);
    constexpr int NPI = (IMGS * (TH + 2) * (TW + 2) + 15) / 16;
    constexpr int NHB = BN == 64 ? 1 : 2;
    static_assert((BN == 64 ? 4 : 2) * (NHB * NPI * 1024 + 4 * BN * 64) <= 160 * 1024, "blocks per CU");
    __shared__ __attribute__((aligned(16))) char lds[NHB * NPI * 1024 + 4 * BN * 64];
    const int SK = p.split_k > 1 ? p.split_k : 1;
    int tile_id, ksplit;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        tile_id = L / SK;
        ksplit = L - tile_id * SK;
    }
    x3m16_item<TW, IMGS, BN, 0>(p, lds, tile_id, ksplit, SK, [] {});
}

#ifdef REID_EXPERIMENTS
// ------------------------------------------------------------------------------------------------------------------------------
// Chain kernel: one persistent launch per ResNet layer for a small batch (a tracking frame).  EXPERIMENT BUILDS ONLY (-DREID_EXPERIMENTS, debug
// switch `chain`): built for layer 4 in round 6, correct at once (4.9e-7 of exact fp32 at 1 / 7 / 30 / 33 crops) and 2.1x SLOWER than the six
// launches it replaces - 536 us against 255 us at 30 crops (profiles/r06_chain_layer4.txt); DESIGN.md section 8 says why.
//
// A 30-crop frame was ~35 launches; every launch is a grid-wide barrier although nothing in the network needs one in eval mode:
// InstanceNorm and SE are per image (SERes18_IBN.py:32-41,88-93), so conv2 of image i waits for conv1 of image i only.  Here the
// layer's work - conv tiles (x3m16_item: same tiles, same split-K reduce-scatter), InstanceNorm finishes, SE tails - is ONE ordered list;
// a block claims the next item with one atomic and waits on the per-image counter of the stage it depends on.  Every item an item can
// wait for comes EARLIER in the list, so it has been claimed by a block that is already running: forward progress needs no
// co-residency assumption and there is no grid barrier.  Producer and consumer sit on different XCDs of one launch: activations go
// through sc1 accesses (COH = 16).  Waits are bounded (fault word bit 2), so a logic error cannot hang the device.
__device__ __forceinline__ void chain_wait(const int* cnt, int target, int* fault) {
    int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > (1 << 22)) {
            if (fault) fault[2] = 1;
            break;
        }
    }
}

__device__ __forceinline__ float chain_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// InstanceNorm finish + [xh | xl'] of the InstanceNorm half of one slice of one image (in_apply_pack_kernel's arithmetic, elementwise.hip)
__device__ __forceinline__ void chain_in_fin(const ChainElem& e, float* sm, int img, int slice, int tid, int* fault) {
    float* sa = sm;
    float* sb = sm + 256;
    for (int ch = tid; ch < e.half; ch += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int t = 0; t < e.tiles; ++t) {
            const float* st = e.stats + (((long long)img * e.tiles + t) * e.c + ch) * 2;
            s1 += (double)chain_ld(st);
            s2 += (double)chain_ld(st + 1);
        }
        const double mean = s1 / e.hw;
        double var = s2 / e.hw - mean * mean;
        if (var < 0.0) var = 0.0;
        const double inv = 1.0 / sqrt(var + 1e-5);
        sa[ch] = (float)(inv * (double)e.g[ch]);
        sb[ch] = (float)((double)e.b[ch] - mean * inv * (double)e.g[ch]);
    }
    __syncthreads();
    const int rows = e.hw / e.slices, q = e.half >> 2;
    const long long pix0 = (long long)img * e.hw + (long long)slice * rows;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(e.x + pix0 * e.c), 0, rows * e.c * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(e.packed + pix0 * 2 * e.c), 0, rows * 2 * e.c * 2, 0x00020000);
    unsigned vm = 0u;
    for (int i = tid; i < rows * q; i += 256) {
        const int row = i / q, cc = i - row * q;
        f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (row * e.c + cc * 4) * 4, 0, 16));
        const f32x4 a = *(const f32x4*)&sa[cc * 4], b = *(const f32x4*)&sb[cc * 4];
        v = v * a + b;
        vm = range_acc(range_acc(range_acc(range_acc(vm, v[0]), v[1]), v[2]), v[3]);
        u32x2 hw, lw;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float v0 = fmaxf(v[2 * h], 0.f), v1 = fmaxf(v[2 * h + 1], 0.f);
            const f16 h0 = (f16)v0, h1 = (f16)v1;
            const f16 l0 = (f16)((v0 - (float)h0) * 2048.0f), l1 = (f16)((v1 - (float)h1) * 2048.0f);
            hw[h] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            lw[h] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
        __builtin_amdgcn_raw_buffer_store_b64(hw, k_rs, (row * 2 * e.c + cc * 4) * 2, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b64(lw, k_rs, (row * 2 * e.c + e.c + cc * 4) * 2, 0, 16);
    }
    range_raise(fault, vm);
}

// SE gate of one image + out = relu(gate * y + shortcut) over one slice of it (se_tail_kernel's arithmetic, elementwise.hip)
__device__ __forceinline__ void chain_se(const ChainElem& e, float* sm, int img, int slice, int tid, int* fault) {
    float* pooled = sm;            // [512]
    float* gate = sm + 512;        // [512]
    float* hid = sm + 1024;        // [64]
    const int lane = tid & 63, wave = tid >> 6, c = e.c, c4v = c >> 2;
    for (int ch = tid; ch < c; ch += 256) {
        double acc = 0.0;
        for (int t = 0; t < e.tiles; ++t) acc += (double)chain_ld(e.stats + (((long long)img * e.tiles + t) * c + ch) * 2);
        pooled[ch] = (float)(acc / e.hw);
    }
    __syncthreads();
    for (int m = wave; m < e.mid; m += 4) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c4 = lane + 64 * k;
            if (c4 < c4v) {
                const f32x4 wv = *(const f32x4*)(e.g + (long long)m * c + c4 * 4), pv = *(const f32x4*)(pooled + c4 * 4);
                acc += wv[0] * pv[0] + wv[1] * pv[1] + wv[2] * pv[2] + wv[3] * pv[3];
            }
        }
        acc = wsum64(acc);
        if (lane == 0) hid[m] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += 256) {
        float acc = 0.f;
        for (int m = 0; m < e.mid; ++m) acc += e.b[m * c + ch] * hid[m];
        gate[ch] = 1.0f / (1.0f + expf(-acc));
    }
    __syncthreads();
    const int rows = e.hw / e.slices;
    const long long pix0 = (long long)img * e.hw + (long long)slice * rows;
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(e.x + pix0 * c), 0, rows * c * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t s_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(e.sc + pix0 * c), 0, rows * c * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)((e.out ? e.out : (float*)e.x) + pix0 * c), 0, rows * c * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t k_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)((e.packed ? e.packed : (f16*)e.x) + pix0 * 2 * c), 0, rows * 2 * c * 2, 0x00020000);
    unsigned vm = 0u;
    for (int i = tid; i < rows * c4v; i += 256) {
        const int row = i / c4v, cc = i - row * c4v;
        const f32x4 yy = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(y_rs, (row * c + cc * 4) * 4, 0, 16));
        const f32x4 rr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s_rs, (row * c + cc * 4) * 4, 0, 16));
        const f32x4 ss = *(const f32x4*)&gate[cc * 4];
        f32x4 o = ss * yy + rr;
        if (e.packed) vm = range_acc(range_acc(range_acc(range_acc(vm, o[0]), o[1]), o[2]), o[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
        if (e.out) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), o_rs, (row * c + cc * 4) * 4, 0, 16);
        if (e.packed) {
            u32x2 hw, lw;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float v0 = o[2 * h], v1 = o[2 * h + 1];
                const f16 h0 = (f16)v0, h1 = (f16)v1;
                const f16 l0 = (f16)((v0 - (float)h0) * 2048.0f), l1 = (f16)((v1 - (float)h1) * 2048.0f);
                hw[h] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                lw[h] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
            }
            __builtin_amdgcn_raw_buffer_store_b64(hw, k_rs, (row * 2 * c + cc * 4) * 2, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b64(lw, k_rs, (row * 2 * c + c + cc * 4) * 2, 0, 16);
        }
    }
    range_raise(fault, vm);
}

template <int TW, int IMGS, int BN>
__global__ __launch_bounds__(256, BN == 64 ? 4 : 2) void chain_kernel(const ChainParams cp) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = 256 / (IMGS * TW);
    constexpr int NPI = (IMGS * (TH + 2) * (TW + 2) + 15) / 16;
    constexpr int NHB = BN == 64 ? 1 : 2;
    constexpr int LDS_BYTES = NHB * NPI * 1024 + 4 * BN * 64;
    static_assert(LDS_BYTES >= 1088 * 4, "the elementwise items keep up to 1088 floats in LDS");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    __shared__ int s_item;
    const int tid = threadIdx.x;
    int* const done = cp.counters + 64;
    for (;;) {
        if (tid == 0) s_item = __hip_atomic_fetch_add(cp.counters, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int item = s_item;
        __syncthreads();                 // (everybody has read the ticket - and has left the previous item's LDS - before either is reused)
        if (item >= cp.total_items) break;
        int s = 0;
        while (s + 1 < cp.n_stages && item >= cp.st[s + 1].first) ++s;
        const ChainStage st = cp.st[s];
        const int li = item - st.first;
        int img_lo, img_hi;              // the images this item reads and finishes: [img_lo, img_hi)
        if (st.kind == 0) {
            const Gemm16Params& p = cp.conv[st.idx];
            const int nnt = p.N / BN;
            const int tile_id = li / st.sk, ksplit = li - tile_id * st.sk, mtile = tile_id / nnt;
            if (IMGS == 2) {
                img_lo = 2 * mtile;
                img_hi = img_lo + 2 < cp.n_img ? img_lo + 2 : cp.n_img;
            } else {
                img_lo = mtile / (p.H / TH);
                img_hi = img_lo + 1;
            }
            auto wait = [&] {
                if (st.dep >= 0) {
                    if (tid == 0)
                        for (int i = img_lo; i < img_hi; ++i) chain_wait(done + st.dep * 64 + i, st.target, cp.fault);
                    __syncthreads();
                }
            };
            if (cp.flags & 4) x3m16_item<TW, IMGS, BN, 0>(p, lds, tile_id, ksplit, st.sk, [] {});      // timing experiment (WRONG results): plain loads, no waits
            else x3m16_item<TW, IMGS, BN, 16>(p, lds, tile_id, ksplit, st.sk, wait);
        } else {
            const ChainElem& e = cp.el[st.idx];
            img_lo = li / e.slices;
            img_hi = img_lo + 1;
            if (st.dep >= 0) {
                if (tid == 0) chain_wait(done + st.dep * 64 + img_lo, st.target, cp.fault);
                __syncthreads();
            }
            if (st.kind == 1) chain_in_fin(e, (float*)lds, img_lo, li - img_lo * e.slices, tid, cp.fault);
            else chain_se(e, (float*)lds, img_lo, li - img_lo * e.slices, tid, cp.fault);
        }
        // the item's stores (sc1: written through) have been acknowledged once vmcnt reaches 0; then its images' counters move
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            for (int i = img_lo; i < img_hi; ++i) __hip_atomic_fetch_add(done + s * 64 + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {      // the last block to leave resets the list for the next launch (stream order: nothing else is in flight then)
        const int old = __hip_atomic_fetch_add(cp.counters + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (int)gridDim.x - 1) {
            for (int i = 0; i < 64 + 8 * 64; ++i) __hip_atomic_store(cp.counters + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#endif
}

#endif   // REID_EXPERIMENTS

// ------------------------------------------------------------------------------------------------------------------------------
// The 128-wide form with the chunk's 27 tiles UNROLLED.  Why: PMC of conv3x3_x3m16_kernel<8,2,128> (profiles/r05_pmc_waits.txt): matrix
// pipe busy 0.51 of the cycles at 2.27 GHz, no LDS bank conflict - and its loop spends ~60 vector + ~80 scalar instructions per tile
// beside 32 MFMAs of 16 cycles each (tap / part decode with a division by three, four A addresses with their swizzle, 64-bit
// source addresses per DMA piece, the halo piece's pixel -> (image, y, x) divisions): as many issue cycles as the MFMAs take.
// With the step index a compile-time constant: the A fragment addresses of the nine taps are 36 registers made once (the halo
// buffer is an instruction offset), DMA pieces go through buffer descriptors with per-lane offsets made once (six halo pieces and
// BJ weight pieces per wave; padding = an offset past the descriptor, which the DMA writes as zeros - as conv_f32.hip does) and
// the step's tap / part / channel offset in the SCALAR operand, waits are immediates.  Same tiles, same order, same arithmetic as
// conv3x3_x3m16_kernel: bit-identical results.
template <class F, int... Rs>
__device__ __forceinline__ void for_each_step(F&& f, std::integer_sequence<int, Rs...>) { (f(std::integral_constant<int, Rs>{}), ...); }

constexpr int x3_halo_step(int r) {                 // does step r (mod 27) of a chunk issue a halo piece?
    r = ((r % 27) + 27) % 27;
    return (r < 6 || (r >= 18 && r < 24)) ? 1 : 0;
}

template <int N>
__device__ __forceinline__ void wait_vm_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TW, int IMGS, int BN, int OCC, int GS = 1>
__global__ __launch_bounds__(256, OCC) void conv3x3_x3u_kernel(const Gemm16Params p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-descriptor builtins exist in the device pass only)
    static_assert((BN == 128 && OCC == 2) || (BN == 64 && (OCC == 3 || OCC == 4)), "128-wide: two blocks per CU, two halo buffers; 64-wide: one halo buffer");
    constexpr int TH = 256 / (IMGS * TW);
    constexpr int WP = TW + 2, HP = TH + 2;
    constexpr int NPX = IMGS * HP * WP;
    constexpr int NPI = (NPX + 15) / 16;
    static_assert(NPI <= 24, "one halo piece per wave and step over six steps");
    constexpr int HALO_BYTES = NPI * 1024;
    constexpr int B_BYTES = BN * 64;
    constexpr int BJ = BN / 64;
    constexpr int TM = 4, TN = BN / 16;
    static_assert(GS == 1 || (GS == 3 && BN == 64), "groups of three steps per barrier: the 64-wide form");
    constexpr int NS = GS == 3 ? 6 : 4;              // weight slots: four single steps in flight, or two groups of three
    constexpr int NHB = BN == 64 ? 1 : 2;
    static_assert(OCC * (NHB * HALO_BYTES + NS * B_BYTES) <= 160 * 1024, "blocks per CU");
    __shared__ __attribute__((aligned(16))) char lds[NHB * HALO_BYTES + NS * B_BYTES];

    // diagnostics (p.diag, tools/x3_stamps.py): s_memtime at entry / loop start / around the single-halo phase waits / loop end / exit
    unsigned long long dg_e = 0, dg_t0 = 0, dg_first = 0, dg_halo = 0, dg_t1 = 0;
#define X3_STAMP(v) if (p.diag) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
    X3_STAMP(dg_e);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int nnt = p.N / BN;
    const int SK = p.split_k > 1 ? p.split_k : 1;
    int mtile, ntile, tile_id, ksplit;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        tile_id = L / SK;
        ksplit = L - tile_id * SK;
        mtile = tile_id / nnt;
        ntile = tile_id - mtile * nnt;
    }
    const int n_blk = ntile * BN;
    const int tiles_per_img = p.H / TH;
    const int img0 = IMGS == 2 ? mtile * 2 : mtile / tiles_per_img;
    const int y0 = IMGS == 2 ? 0 : (mtile - img0 * tiles_per_img) * TH;
    const int n_img = p.M / (p.H * p.W);
    const int C = p.Cin / 3;
    const int a_cin = 2 * C;
    const int ncr_all = C / 32;
    const int ncr = ncr_all / SK, c0 = ksplit * ncr;
    const unsigned halo32 = (unsigned)(uintptr_t)lds, ring32 = halo32 + NHB * HALO_BYTES;

    // ---- descriptors and per-lane offsets, made once
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((long long)n_img * p.H * p.W * a_cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((long long)p.N * p.ldb * 2), 0x00020000);
    int hv[6];                                        // this wave's halo pieces k = 0 .. 5: piece q = 4 k + wave (the surplus ones repeat the last)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int q = k * 4 + wm < NPI ? k * 4 + wm : NPI - 1;
        const int hp = q * 16 + (lane >> 2);
        const int im = hp / (HP * WP), rem = hp - im * (HP * WP);
        const int hy = rem / WP, hx = rem - hy * WP;
        const int gy = y0 - 1 + hy, gx = hx - 1, gi = img0 + im;
        const bool ok = hp < NPX && gi < n_img && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const int cg = (lane & 3) ^ (((hp >> 3) & 1) << 1);
        hv[k] = ok ? (int)(((((long long)gi * p.H + gy) * p.W + gx) * a_cin + cg * 8) * 2) : 0x7fffff00;   // past the descriptor: zeros
    }
    int wv[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = (wm * BJ + j) * 16 + (lane >> 2);
        const int cg = (lane & 3) ^ (((row >> 3) & 1) << 1);
        wv[j] = (int)((((long long)(n_blk + row) * p.ldb) + cg * 8) * 2);
    }
    const int cin2 = p.Cin * 2, c2 = C * 2;           // bytes per tap / per weight part of a weight row
    auto issue_w = [&](int cc, int r, int slot) __attribute__((always_inline)) {   // weight tile of step r of chunk cc (r compile-time at the call sites)
        const int part = r < 9 ? 0 : r < 18 ? 2 : 1;
        const int tap = r < 9 ? r : r < 18 ? r - 9 : r - 18;
        const int soff = tap * cin2 + part * c2 + (c0 + cc) * 64;
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, LPTR(lds + NHB * HALO_BYTES + slot * B_BYTES + (wm * BJ + j) * 1024), 16, wv[j], soff, 0, 0);
    };
    auto issue_halo = [&](int k, int vchunk, int buf) __attribute__((always_inline)) {
        const int q = k * 4 + wm < NPI ? k * 4 + wm : NPI - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, LPTR(lds + buf * HALO_BYTES + q * 1024), 16, hv[k], vchunk * 64, 0, 0);
    };

    // ---- prologue, issued BEFORE the fragment addresses below are made (~3 k cycles of setup beside the other blocks' MFMAs:
    // s_memtime stamps, round 5 - the first loads now fly under it): xh of the first chunk and the weight tiles of steps 0 .. 2
#pragma unroll
    for (int k = 0; k < 6; ++k) issue_halo(k, c0, 0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    issue_w(0, 2, 2);

    const int pj = pi16(l16);
    unsigned aa[9][TM];                               // A fragment address of (tap, 16-row tile) in halo buffer 0
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        int h0;
        if constexpr (TW == 32) h0 = (2 * wm + (a >> 1)) * WP + 16 * (a & 1) + pj;
        else if constexpr (TW == 16) h0 = (4 * wm + a) * WP + pj;
        else h0 = (wm >> 1) * HP * WP + ((wm & 1) * 8 + a + 4 * (pj >> 3)) * WP + (pj & 7);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int hp = h0 + (tap / 3) * WP + tap % 3;
            aa[tap][a] = halo32 + hp * 64 + ((lq ^ (((hp >> 3) & 1) << 1)) * 16);
        }
    }
    const unsigned bx = ring32 + pj * 64 + ((lq ^ (((pj >> 3) & 1) << 1)) * 16);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    X3_STAMP(dg_t0);
    for (int c = 0; c < ncr; ++c) {
        const int tb = c * 27;
        const int gpar = (c * 9) & 1;                // GS == 3: parity of the chunk's first group (a chunk is nine groups)
        const bool last = c + 1 == ncr;
        for_each_step([&](auto RC) __attribute__((always_inline)) {
            constexpr int r = decltype(RC)::value;
            if constexpr (NHB == 2) {
                // in-order landing: all but the pieces of the last two steps have landed (conv3x3_x3m16_kernel)
                constexpr int allow = 2 * BJ + x3_halo_step(r - 1) + x3_halo_step(r - 2);
                if constexpr (r + 2 >= 27) {
                    if (last) wait_vm_imm<0>(); else wait_vm_imm<allow>();
                } else {
                    wait_vm_imm<allow>();
                }
                RAW_BARRIER();
                if constexpr (r + 3 < 27) issue_w(c, r + 3, (tb + r + 3) & 3);
                else if (!last) issue_w(c + 1, r + 3 - 27, (tb + r + 3) & 3);
                if constexpr (r < 6) issue_halo(r, ncr_all + c0 + c, 1);
                if constexpr (r >= 18 && r < 24) issue_halo(r - 18, c0 + (last ? c : c + 1), 0);
            } else if constexpr (GS == 1) {
                // one halo buffer: a phase (xh: steps 0 .. 17, xl': 18 .. 26) starts with a barrier (every wave has read the old halo),
                // this wave's six pieces and vmcnt(0); inside a phase only the weight ring is in flight
                constexpr bool phase_start = r == 0 || r == 18;
                if constexpr (phase_start) {
                    unsigned long long ta = 0, tb2 = 0;
                    X3_STAMP(ta);
                    if (r == 18 || c > 0) {            // (the first chunk's xh: requested in the prologue)
                        RAW_BARRIER();
#pragma unroll
                        for (int k = 0; k < 6; ++k) issue_halo(k, r == 0 ? c0 + c : ncr_all + c0 + c, 0);
                    }
                    wait_vm_imm<0>();
                    if (p.diag) {
                        X3_STAMP(tb2);
                        if (r == 0 && c == 0) dg_first = tb2 - ta; else dg_halo += tb2 - ta;
                    }
                } else if constexpr (r + 2 >= 27) {
                    if (last) wait_vm_imm<0>(); else wait_vm_imm<2 * BJ>();
                } else {
                    wait_vm_imm<2 * BJ>();
                }
                RAW_BARRIER();
                if constexpr (r + 3 < 27) issue_w(c, r + 3, (tb + r + 3) & 3);
                else if (!last) issue_w(c + 1, r + 3 - 27, (tb + r + 3) & 3);
            } else {
                // groups of three steps per barrier (a 64-wide step is 16 MFMAs per wave: 256 cycles between two barriers): at a
                // group's start the group's weights - requested one group ago - and, at a phase's start, the halo have landed
                // (vmcnt(0): nothing newer is in flight), one barrier, then the NEXT group's three weight tiles are requested
                if constexpr (r % 3 == 0) {
                    constexpr bool phase_start = r == 0 || r == 18;
                    if constexpr (phase_start) {
                        if (r == 18 || c > 0) {
                            RAW_BARRIER();
#pragma unroll
                            for (int k = 0; k < 6; ++k) issue_halo(k, r == 0 ? c0 + c : ncr_all + c0 + c, 0);
                        }
                    }
                    wait_vm_imm<0>();
                    RAW_BARRIER();
                    const int nslot = (((gpar + r / 3) & 1) ^ 1) * 3;
                    if constexpr (r + 3 < 27) {
                        issue_w(c, r + 3, nslot); issue_w(c, r + 4, nslot + 1); issue_w(c, r + 5, nslot + 2);
                    } else if (!last) {
                        issue_w(c + 1, 0, nslot); issue_w(c + 1, 1, nslot + 1); issue_w(c + 1, 2, nslot + 2);
                    }
                }
            }
            constexpr int tap = r < 9 ? r : r < 18 ? r - 9 : r - 18;
            constexpr int hoff = (NHB == 2 && r >= 18) ? HALO_BYTES : 0;
            const unsigned ba = bx + (unsigned)((GS == 3 ? ((gpar + r / 3) & 1) * 3 + r % 3 : (tb + r) & 3) * B_BYTES);
            half8 fa[TM], fb[TN];
            LDS_READ(fa[0], aa[tap][0], hoff);
            LDS_READ(fb[0], ba, 0);
            LDS_READ(fb[1], ba, 1024);
            LDS_READ(fb[2], ba, 2048);
            LDS_READ(fb[3], ba, 3072);
            if constexpr (TN == 8) {
                LDS_READ(fb[4], ba, 4096);
                LDS_READ(fb[5], ba, 5120);
                LDS_READ(fb[6], ba, 6144);
                LDS_READ(fb[7], ba, 7168);
            }
            LDS_READ(fa[1], aa[tap][1], hoff);
            LDS_READ(fa[2], aa[tap][2], hoff);
            LDS_READ(fa[3], aa[tap][3], hoff);
            constexpr int NRD = TM + TN;
#define MMA(a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[a][b]) : "v"(fb[b]), "v"(fa[a]))   /* weights first: D^T (x3m16_tail) */
#define MM0(b)                                   \
    do {                                         \
        lgkm_wait1<NRD - 2 - (b)>(fb[b]);        \
        MMA(0, b);                               \
    } while (0)
            lgkm_wait1<NRD - 1>(fa[0]);
            MM0(0); MM0(1); MM0(2); MM0(3);
            if constexpr (TN == 8) { MM0(4); MM0(5); MM0(6); MM0(7); }
            lgkm_wait1<2>(fa[1]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(1, b);
            lgkm_wait1<1>(fa[2]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(2, b);
            lgkm_wait1<0>(fa[3]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(3, b);
#undef MM0
#undef MMA
        }, std::make_integer_sequence<int, 27>{});
    }
    X3_STAMP(dg_t1);
    x3m16_tail<TW, IMGS, BN>(p, acc, lds, tid, wm, mtile, n_blk, tile_id, ksplit, SK);
    if (p.diag) {   // blocks from the middle of the launch, per wave: set-up | first halo wait | later halo waits | entry -> loop end | epilogue
        unsigned long long dg_t2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        X3_STAMP(dg_t2);
        const int bi = (int)blockIdx.x - (int)(gridDim.x / 2);
        if (bi >= 0 && bi < 64 && lane == 0) {
            unsigned long long* o = p.diag + ((size_t)bi * 8 + wm) * 5;
            o[0] = dg_t0 - dg_e; o[1] = dg_first; o[2] = dg_halo; o[3] = dg_t1 - dg_e; o[4] = dg_t2 - dg_t1;
        }
    }
#undef X3_STAMP
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same block structure for a DENSE contraction (the Swin linears of stages 3-4 in the fp32-class mode, swin_transformer.py:23-39,
// 191-232): C[M][N] = A[M][K] . W[N][K]^T with A = [xh | xl'] f16 [M][2 K] and W = [wh 2^11 | wh | wl'] f16 [N][3 K].
// 256 x 128 tile, four waves (a wave: 64 rows x 128 columns, 128 accumulator registers), TWO blocks per CU, v_mfma_f32_16x16x32_f16,
// a real 32-channel chunk = three steps: xh.wh 2^11, xh.wl' (the xh fragments stay in registers), xl'.wh.  There is no halo to re-use:
// a chunk moves two 16-KB A parts and three 8-KB weight tiles, 14 DMA pieces per wave and chunk (gemm_f16.hip's K loop over the virtual
// 3 K columns: 18, xh twice).  LDS: THREE A slots through which the parts h(0) l(0) h(1) l(1) ... rotate (part p in slot p % 3; part
// p + 3 is requested at the barrier behind part p's last use) and FOUR weight slots (step s in slot s & 3), both three steps ahead;
// counted vmcnt (in-order landing): at a step the pieces of the last two steps may still be in flight.
// Per step: r = 0 requests weights(s + 3) and h(c + 1) (6 pieces per wave), r = 1 weights (2), r = 2 weights and l(c + 1) (6).
// Ragged M: rows past M are an offset past the A descriptor (zeros) and a store offset past the output descriptors (dropped).
template <int BN>
__global__ __launch_bounds__(256, 2) void lin_x3_kernel(const Gemm16Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TM = 4, TN = BN / 16, BJ = BN / 64;
    constexpr int A_SLOT = 256 * 64, B_SLOT = BN * 64;
    static_assert(2 * (3 * A_SLOT + 4 * B_SLOT) <= 160 * 1024, "two blocks per CU");
    __shared__ __attribute__((aligned(16))) char lds[3 * A_SLOT + 4 * B_SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int nnt = p.N / BN;
    int mtile, ntile;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        mtile = L / nnt;
        ntile = L - mtile * nnt;
    }
    const int m_blk = mtile * 256, n_blk = ntile * BN;
    const int m_valid = p.M - m_blk < 256 ? p.M - m_blk : 256;
    const int Kr = p.K / 3;                       // real K; A rows are [xh (Kr) | xl' (Kr)], weight rows [wh 2^11 | wh | wl'] (Kr each)
    const int ncr = Kr / 32;
    const unsigned a32 = (unsigned)(uintptr_t)lds, b32 = a32 + 3 * A_SLOT;

    const __amdgpu_buffer_rsrc_t a_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long long)m_blk * p.lda), 0, (int)((long long)m_valid * p.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (long long)n_blk * p.ldb), 0, (int)((long long)BN * p.ldb * 2), 0x00020000);
    int av[4], wv[BJ];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                 // this wave's A pieces: rows (4 wm + j) 16 .. + 16 of the tile
        const int row = (wm * 4 + j) * 16 + (lane >> 2);
        const int cg = (lane & 3) ^ (((row >> 3) & 1) << 1);
        av[j] = row < m_valid ? (int)(((long long)row * p.lda + cg * 8) * 2) : 0x7fffff00;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = (wm * BJ + j) * 16 + (lane >> 2);
        const int cg = (lane & 3) ^ (((row >> 3) & 1) << 1);
        wv[j] = (int)(((long long)row * p.ldb + cg * 8) * 2);
    }
    const int abl = p.ablate;     // timing experiments (debug switch x3_ablate; WRONG results): 1 no weight DMA in the loop, 2 no A DMA, 4 no fragment reads
    auto issue_a = [&](int part, int slot) __attribute__((always_inline)) {      // part 2 c: xh of chunk c, 2 c + 1: xl'
        if ((abl & 2) && part > 1) return;
        const int soff = (part & 1) * Kr * 2 + (part >> 1) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, LPTR(lds + slot * A_SLOT + (wm * 4 + j) * 1024), 16, av[j], soff, 0, 0);
    };
    // Round 6: wh is streamed ONCE per chunk.  The rows [wh 2^11 | wh | wl'] of the weight image serve three products per real chunk -
    // xh.(wh 2^11), xh.wl', xl'.wh - and the first and third want the same 16 bits up to an exponent: the wh fragments stay in registers
    // from step 0 to step 2 and step 0 multiplies them by 2^11 on the way into its MFMAs (v_pk_mul_f16, exact: the load-time range check
    // guarantees |w| 2^11 inside f16, and the stored [wh 2^11] third IS f16(wh) 2^11).  Two weight tiles per chunk instead of three (4 rows
    // per 32 channels instead of 5: -14 % of the LDS-DMA bytes this kernel is bound by, -8 of 24 fragment reads); same three MFMAs per
    // accumulator in the same order: bit-identical results.
    auto issue_w = [&](int c, int kind, int slot) __attribute__((always_inline)) {  // kind 0: the wh tile of chunk c, 1: the wl' tile
        if ((abl & 1) && c > 0) return;
        const int part = kind == 0 ? 1 : 2;
        const int soff = part * Kr * 2 + c * 64;
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, LPTR(lds + 3 * A_SLOT + slot * B_SLOT + (wm * BJ + j) * 1024), 16, wv[j], soff, 0, 0);
    };
    // prologue = chunk 0's requests in the order every later chunk's are made: wh, h, wl', l
    issue_w(0, 0, 0);
    issue_a(0, 0);
    issue_w(0, 1, 1);
    issue_a(1, 1);

    const int pj = pi16(l16);
    const unsigned swz = (unsigned)((lq ^ (((pj >> 3) & 1) << 1)) * 16);
    unsigned aa[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) aa[a] = a32 + (unsigned)((wm * 64 + a * 16 + pj) * 64) + swz;
    const unsigned bx = b32 + (unsigned)(pj * 64) + swz;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    half8 fa[TM], fb[TN], fw[TN];                 // fw: the chunk's wh fragments, alive from step 0 to step 2
    int sa_h = 0, sa_l = 1;                       // A slots of this chunk's parts: (2 c) % 3, (2 c + 1) % 3
    const half8 k2048 = {(f16)2048.0f, (f16)2048.0f, (f16)2048.0f, (f16)2048.0f, (f16)2048.0f, (f16)2048.0f, (f16)2048.0f, (f16)2048.0f};
    for (int c = 0; c < ncr; ++c) {
        const bool last = c + 1 == ncr;
        const int s0 = c * 2;                     // weight tiles 2 c (wh) and 2 c + 1 (wl') in ring slots (tile & 3)
#define MMA(a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[a][b]) : "v"(fb[b]), "v"(fa[a]))   /* weights first: D^T (x3m16_tail) */
#define MMAW(a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[a][b]) : "v"(fw[b]), "v"(fa[a]))
#define LIN_B_READS(f, ba)                              \
    LDS_READ(f[0], ba, 0);     LDS_READ(f[1], ba, 1024); \
    LDS_READ(f[2], ba, 2048);  LDS_READ(f[3], ba, 3072); \
    LDS_READ(f[4], ba, 4096);  LDS_READ(f[5], ba, 5120); \
    LDS_READ(f[6], ba, 6144);  LDS_READ(f[7], ba, 7168)
        static_assert(TN == 8, "fragment reads below are written for the 128-wide tile");
        // Requests of a chunk, in order: [wh(c+1) + h(c+1)] at step 0, [wl'(c+1)] at step 1, [l(c+1)] at step 2.  In-order landing: a
        // step waits until at most the requests made AFTER the ones it needs are in flight.
        {   // ---- step 0: xh . (wh 2^11); needs wh(c), h(c); behind them: wl'(c) [BJ], l(c) [4]
            wait_vm_imm<BJ + 4>();
            RAW_BARRIER();
            if (!last) {
                issue_w(c + 1, 0, (s0 + 2) & 3);
                issue_a(2 * c + 2, sa_h == 0 ? 2 : sa_h - 1);     // h(c + 1) -> slot (2 c + 2) % 3 = (sa_h + 2) % 3
            }
            const unsigned ao = (unsigned)(sa_h * A_SLOT), ba = bx + (unsigned)((s0 & 3) * B_SLOT);
            if (!(abl & 4)) {
                LDS_READ(fa[0], aa[0] + ao, 0);
                LDS_READ(fa[1], aa[1] + ao, 0);
                LDS_READ(fa[2], aa[2] + ao, 0);
                LDS_READ(fa[3], aa[3] + ao, 0);
                LIN_B_READS(fw, ba);
            }
            // column unit by column unit: the scaled copy of a wh fragment lives in fb[b] for its four MFMAs only
#define STEP0(b, n)                                                                  \
    do {                                                                             \
        lgkm_wait1<n>(fw[b]);                                                        \
        fb[b] = fw[b] * k2048;                                                       \
        /* the MFMAs are inline asm: hipcc does not see that they read what the v_pk_mul above has just written (VALU write -> MFMA */ \
        /* operand read needs wait states; without them the MFMAs took stale registers: inf / NaN) */ \
        asm volatile("s_nop 3" : "+v"(fb[b]));                                       \
        MMA(0, b); MMA(1, b); MMA(2, b); MMA(3, b);                                  \
    } while (0)
            lgkm_wait1<8>(fa[3]);                 // (reads return in order: the four A fragments are in)
            asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]));
            STEP0(0, 7); STEP0(1, 6); STEP0(2, 5); STEP0(3, 4); STEP0(4, 3); STEP0(5, 2); STEP0(6, 1); STEP0(7, 0);
#undef STEP0
        }
        {   // ---- step 1: xh . wl' (the xh fragments are in registers); needs wl'(c); behind it: l(c) [4], wh(c+1) + h(c+1) [BJ + 4]
            if (last) wait_vm_imm<4>(); else wait_vm_imm<BJ + 8>();
            RAW_BARRIER();
            if (!last) issue_w(c + 1, 1, (s0 + 3) & 3);
            const unsigned ba = bx + (unsigned)(((s0 + 1) & 3) * B_SLOT);
            if (!(abl & 4)) { LIN_B_READS(fb, ba); }
            lgkm_wait1<7>(fb[0]); MMA(0, 0);
            lgkm_wait1<6>(fb[1]); MMA(0, 1);
            lgkm_wait1<5>(fb[2]); MMA(0, 2);
            lgkm_wait1<4>(fb[3]); MMA(0, 3);
            lgkm_wait1<3>(fb[4]); MMA(0, 4);
            lgkm_wait1<2>(fb[5]); MMA(0, 5);
            lgkm_wait1<1>(fb[6]); MMA(0, 6);
            lgkm_wait1<0>(fb[7]); MMA(0, 7);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(1, b);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(2, b);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(3, b);
        }
        {   // ---- step 2: xl' . wh (the wh fragments are in registers); needs l(c); behind it: wh(c+1) + h(c+1) [BJ + 4], wl'(c+1) [BJ]
            if (last) wait_vm_imm<0>(); else wait_vm_imm<2 * BJ + 4>();
            RAW_BARRIER();
            if (!last) issue_a(2 * c + 3, sa_h);                   // l(c + 1) -> slot (2 c + 3) % 3 = the slot h(c) just left
            const unsigned ao = (unsigned)(sa_l * A_SLOT);
            if (!(abl & 4)) {
                LDS_READ(fa[0], aa[0] + ao, 0);
                LDS_READ(fa[1], aa[1] + ao, 0);
                LDS_READ(fa[2], aa[2] + ao, 0);
                LDS_READ(fa[3], aa[3] + ao, 0);
            }
            lgkm_wait1<3>(fa[0]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMAW(0, b);
            lgkm_wait1<2>(fa[1]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMAW(1, b);
            lgkm_wait1<1>(fa[2]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMAW(2, b);
            lgkm_wait1<0>(fa[3]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMAW(3, b);
        }
#undef LIN_B_READS
#undef MMAW
#undef MMA
        // next chunk: h -> (2 c + 2) % 3, l -> (2 c + 3) % 3
        const int nh = sa_h == 0 ? 2 : sa_h - 1;
        sa_l = sa_h;
        sa_h = nh;
    }
    x3m16_tail<16, 1, BN, true>(p, acc, lds, tid, wm, mtile, n_blk, 0, 0, 1);
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------
// Strided 3x3 and 1x1 convolutions of the fp32-class mode (SERes18_IBN.py:120-128 conv1 of a down-sampling block, :250-276 the shortcut
// convolutions) - round 6.  They have no halo a block could keep: an output pixel's taps are other pixels' taps only every second
// column.  Until now they ran as gemm_f16.hip's SPLIT im2col build (12 waves, 32x32x16 MFMAs, no split-K form: pipe busy 0.245 at 1024
// crops) and, for tracking-sized batches, on the EXACT-fp32 kernel (a layer's arithmetic depended on the batch size; 5 of a 30-crop
// frame's 35 launches, 103 us).  Here: lin_x3_kernel's block - 256 x 128 tile, four waves, two blocks per CU, three A slots and four
// weight slots three steps ahead, wh fragments kept for the third product - with the A rows GATHERED: row m of the tile is output pixel
// (img, oy, ox), step (tap, chunk) reads 64 bytes of input pixel (oy st + r - pad, ox st + s - pad) of the packed [xh | xl'] image; the
// per-lane part of the address is the pixel of tap (0, 0), the tap and chunk offsets are scalar, a tap outside the image is an offset
// past the descriptor (zeros).  K order: taps outer, 32-channel chunks inner, the three products of a chunk in lin_x3's order.
// Split-K over the (tap, chunk) steps with x3m16_tail's reduce-scatter; the convolution epilogue (folded BatchNorm, ReLU from
// relu_from, column sums, [yh | yl'] store from pack_from) is x3m16_tail<16, 1, BN, false>: rows in natural NHWC order.
template <int BN>
__global__ __launch_bounds__(256, 2) void conv_x3s_kernel(const Gemm16Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TM = 4, TN = BN / 16, BJ = BN / 64;
    constexpr int A_SLOT = 256 * 64, B_SLOT = BN * 64;
    static_assert(2 * (3 * A_SLOT + 4 * B_SLOT) <= 160 * 1024, "two blocks per CU");
    static_assert(TN == 8, "fragment reads below are written for the 128-wide tile");
    __shared__ __attribute__((aligned(16))) char lds[3 * A_SLOT + 4 * B_SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int nnt = p.N / BN;
    const int SK = p.split_k > 1 ? p.split_k : 1;
    int mtile, ntile, tile_id, ksplit;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        tile_id = L / SK;
        ksplit = L - tile_id * SK;
        mtile = tile_id / nnt;
        ntile = tile_id - mtile * nnt;
    }
    const int m_blk = mtile * 256, n_blk = ntile * BN;
    const int m_valid = p.M - m_blk < 256 ? p.M - m_blk : 256;
    const int C = p.Cin / 3;                      // real input channels; a pixel of A is [xh (C) | xl' (C)], a tap of a weight row [wh 2^11 | wh | wl'] (C each)
    const int ncr = C / 32;
    const int nq = p.R * p.S * ncr;               // (tap, chunk) steps of the whole K loop; this block: [q0, q1)
    const int q0 = ksplit * nq / SK, q1 = (ksplit + 1) * nq / SK;      // (nq need not divide: shares differ by one step)
    const int pix_b = 2 * C * 2;                  // bytes of an input pixel
    const unsigned a32 = (unsigned)(uintptr_t)lds, b32 = a32 + 3 * A_SLOT;

    const int n_img = p.M / (p.Ho * p.Wo);
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((long long)n_img * p.H * p.W * pix_b), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (long long)n_blk * p.ldb), 0, (int)((long long)BN * p.ldb * 2), 0x00020000);
    int av[4], rmask[4], cmask[4], wv[BJ];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                 // this wave's A pieces: rows (4 wm + j) 16 .. + 16 of the tile, four lanes per row
        const int row = (wm * 4 + j) * 16 + (lane >> 2);
        const int cg = (lane & 3) ^ (((row >> 3) & 1) << 1);
        const int m = m_blk + row;
        const int img = m / (p.Ho * p.Wo), rem = m - img * (p.Ho * p.Wo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        av[j] = ((img * p.H + iy0) * p.W + ix0) * pix_b + cg * 16;          // tap (0, 0); may lie outside (then its mask bit is clear)
        int rm = 0, cm = 0;
        if (row < m_valid) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                rm |= ((unsigned)(iy0 + t) < (unsigned)p.H && t < p.R) ? 1 << t : 0;
                cm |= ((unsigned)(ix0 + t) < (unsigned)p.W && t < p.S) ? 1 << t : 0;
            }
        }
        rmask[j] = rm; cmask[j] = cm;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = (wm * BJ + j) * 16 + (lane >> 2);
        const int cg = (lane & 3) ^ (((row >> 3) & 1) << 1);
        wv[j] = (int)(((long long)row * p.ldb + cg * 8) * 2);
    }
    // step q = (tap t = (r, s), chunk c): scalar state of the NEXT request, advanced as requests are made (A parts and weight tiles of a
    // step are requested in the order wh, h, wl', l as in lin_x3_kernel, so one running (t, r, s, c) serves all four)
    auto issue_a = [&](int r, int s, int c, int part, int slot) __attribute__((always_inline)) {      // part 0: xh, 1: xl'
        const int tap_b = (r * p.W + s) * pix_b, soff = part * C * 2 + c * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = ((rmask[j] >> r) & 1) && ((cmask[j] >> s) & 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, LPTR(lds + slot * A_SLOT + (wm * 4 + j) * 1024), 16, in ? av[j] + tap_b : 0x7fffff00, soff, 0, 0);
        }
    };
    auto issue_w = [&](int t, int c, int kind, int slot) __attribute__((always_inline)) {   // kind 0: the wh tile, 1: the wl' tile
        const int soff = (t * 3 + (kind == 0 ? 1 : 2)) * C * 2 + c * 64;
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, LPTR(lds + 3 * A_SLOT + slot * B_SLOT + (wm * BJ + j) * 1024), 16, wv[j], soff, 0, 0);
    };
    int nt = q0 / ncr, nc = q0 - nt * ncr, nr = nt / p.S, ns = nt - nr * p.S;        // the step whose operands are requested next
    auto advance = [&]() __attribute__((always_inline)) {
        if (++nc == ncr) {
            nc = 0; ++nt;
            if (++ns == p.S) { ns = 0; ++nr; }
        }
    };
    issue_w(nt, nc, 0, 0);
    issue_a(nr, ns, nc, 0, 0);
    issue_w(nt, nc, 1, 1);
    issue_a(nr, ns, nc, 1, 1);
    advance();

    const int pj = pi16(l16);
    const unsigned swz = (unsigned)((lq ^ (((pj >> 3) & 1) << 1)) * 16);
    unsigned aa[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) aa[a] = a32 + (unsigned)((wm * 64 + a * 16 + pj) * 64) + swz;
    const unsigned bx = b32 + (unsigned)(pj * 64) + swz;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    half8 fa[TM], fb[TN], fw[TN];
    int sa_h = 0, sa_l = 1;
    const half8 k2048 = {(f16)2048.0f, (f16)2048.0f, (f16)2048.0f, (f16)2048.0f, (f16)2048.0f, (f16)2048.0f, (f16)2048.0f, (f16)2048.0f};
    for (int q = q0; q < q1; ++q) {
        const bool last = q + 1 == q1;
        const int s0 = (q - q0) * 2;              // weight tiles 2 i (wh) and 2 i + 1 (wl') of this block's step i in ring slots (tile & 3)
#define MMA(a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[a][b]) : "v"(fb[b]), "v"(fa[a]))   /* weights first: D^T (x3m16_tail) */
#define MMAW(a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[a][b]) : "v"(fw[b]), "v"(fa[a]))
#define LIN_B_READS(f, ba)                              \
    LDS_READ(f[0], ba, 0);     LDS_READ(f[1], ba, 1024); \
    LDS_READ(f[2], ba, 2048);  LDS_READ(f[3], ba, 3072); \
    LDS_READ(f[4], ba, 4096);  LDS_READ(f[5], ba, 5120); \
    LDS_READ(f[6], ba, 6144);  LDS_READ(f[7], ba, 7168)
        {   // ---- product 0: xh . (wh 2^11); needs wh(q), h(q); behind them: wl'(q) [BJ], l(q) [4]
            wait_vm_imm<BJ + 4>();
            RAW_BARRIER();
            if (!last) {
                issue_w(nt, nc, 0, (s0 + 2) & 3);
                issue_a(nr, ns, nc, 0, sa_h == 0 ? 2 : sa_h - 1);
            }
            const unsigned ao = (unsigned)(sa_h * A_SLOT), ba = bx + (unsigned)((s0 & 3) * B_SLOT);
            LDS_READ(fa[0], aa[0] + ao, 0);
            LDS_READ(fa[1], aa[1] + ao, 0);
            LDS_READ(fa[2], aa[2] + ao, 0);
            LDS_READ(fa[3], aa[3] + ao, 0);
            LIN_B_READS(fw, ba);
#define STEP0(b, n)                                                                  \
    do {                                                                             \
        lgkm_wait1<n>(fw[b]);                                                        \
        fb[b] = fw[b] * k2048;                                                       \
        asm volatile("s_nop 3" : "+v"(fb[b]));   /* VALU write -> inline-asm MFMA read: wait states the compiler does not count (lin_x3_kernel) */ \
        MMA(0, b); MMA(1, b); MMA(2, b); MMA(3, b);                                  \
    } while (0)
            lgkm_wait1<8>(fa[3]);
            asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]));
            STEP0(0, 7); STEP0(1, 6); STEP0(2, 5); STEP0(3, 4); STEP0(4, 3); STEP0(5, 2); STEP0(6, 1); STEP0(7, 0);
#undef STEP0
        }
        {   // ---- product 1: xh . wl'; needs wl'(q); behind it: l(q) [4], wh(q+1) + h(q+1) [BJ + 4]
            if (last) wait_vm_imm<4>(); else wait_vm_imm<BJ + 8>();
            RAW_BARRIER();
            if (!last) issue_w(nt, nc, 1, (s0 + 3) & 3);
            const unsigned ba = bx + (unsigned)(((s0 + 1) & 3) * B_SLOT);
            LIN_B_READS(fb, ba);
            lgkm_wait1<7>(fb[0]); MMA(0, 0);
            lgkm_wait1<6>(fb[1]); MMA(0, 1);
            lgkm_wait1<5>(fb[2]); MMA(0, 2);
            lgkm_wait1<4>(fb[3]); MMA(0, 3);
            lgkm_wait1<3>(fb[4]); MMA(0, 4);
            lgkm_wait1<2>(fb[5]); MMA(0, 5);
            lgkm_wait1<1>(fb[6]); MMA(0, 6);
            lgkm_wait1<0>(fb[7]); MMA(0, 7);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(1, b);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(2, b);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMA(3, b);
        }
        {   // ---- product 2: xl' . wh (the wh fragments are in registers); needs l(q); behind it: wh(q+1) + h(q+1) [BJ + 4], wl'(q+1) [BJ]
            if (last) wait_vm_imm<0>(); else wait_vm_imm<2 * BJ + 4>();
            RAW_BARRIER();
            if (!last) {
                issue_a(nr, ns, nc, 1, sa_h);
                advance();
            }
            const unsigned ao = (unsigned)(sa_l * A_SLOT);
            LDS_READ(fa[0], aa[0] + ao, 0);
            LDS_READ(fa[1], aa[1] + ao, 0);
            LDS_READ(fa[2], aa[2] + ao, 0);
            LDS_READ(fa[3], aa[3] + ao, 0);
            lgkm_wait1<3>(fa[0]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMAW(0, b);
            lgkm_wait1<2>(fa[1]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMAW(1, b);
            lgkm_wait1<1>(fa[2]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMAW(2, b);
            lgkm_wait1<0>(fa[3]);
#pragma unroll
            for (int b = 0; b < TN; ++b) MMAW(3, b);
        }
#undef LIN_B_READS
#undef MMAW
#undef MMA
        const int nh = sa_h == 0 ? 2 : sa_h - 1;
        sa_l = sa_h;
        sa_h = nh;
    }
    x3m16_tail<16, 1, BN, false>(p, acc, lds, tid, wm, mtile, n_blk, tile_id, ksplit, SK);
#endif
}

// few output tiles (a tracking frame, a camera group): the real 32-channel chunks split over sk blocks per tile, up to two blocks for
// every CU (x3m16_tail hands the tile's sixteen-column units out to the sk blocks: sk divides their number).  Eight ways lose to four -
// twice the partial traffic: 30 crops 882 -> 1175 us per pass - so four is the cap.
static int x3_split(const reid_ctx* ctx, int tiles, int ncr, bool wide, int slots = 512) {
    int sk = 1;
    while (ctx->f16_split_k && tiles * sk * 2 <= slots && ncr % (sk * 2) == 0 && sk * 2 <= (wide ? 8 : 4)) sk *= 2;
    const int cap = ctx->x3_sk_cap > 0 ? ctx->x3_sk_cap : 4;
    return sk > cap ? cap : sk;
}

// 128- or 64-wide tiles.  64-wide (three to four blocks per CU): layer 1 always; switch x3_narrow: bit 0 = also the 16-wide maps (layer 2:
// default), bit 1 = every 8-wide one.  Round 6: layer 3 (8-wide maps, 256 channels) takes the 64-wide tile when the 128-wide launch would
// leave the chip half empty (fewer than 256 blocks: up to ~60 crops) - measured (gpurun_out/r6/timeline_*_nar*, timeline_30_allx3n3): 30 crops 31
// (12-wave) -> 23 us per convolution, 48: 36 -> 33; at 64 crops (both forms four ways split) the wide tile stays (40 against 44).  Layer 4
// loses with narrow tiles from 120 crops on and keeps the wide one.
static bool x3_wide_tiles(const reid_ctx* ctx, const Gemm16Params& p) {
    if (p.N % 128 != 0 || (p.W == 16 && (ctx->x3_narrow & 1)) || (p.W == 8 && (ctx->x3_narrow & 2))) return false;
    if (p.W == 8 && p.N == 256 && !(ctx->x3_narrow & 4)) {       // (bit 2: this rule off)
        const int nmt = (p.M + 255) / 256, ncr = p.Cin / 3 / 32;
        const int t128 = nmt * 2, t64 = nmt * 4;
        const int s128 = x3_split(ctx, t128, ncr, true);
        (void)t64;
        // (the narrow tile also wins where it needs fewer splits or runs unsplit below 512 blocks - 120 / 200 / 256 / 384 crops: 1 % of a pass -
        // but there it changes the summation order of passes the config-1 rank vectors are checked at; not worth another sub-noise row)
        if (t128 * s128 < 256) return false;
    }
    // Layer 4 (8-wide maps of image pairs, 512 channels), round 6 - pass sizes where the 128-wide launch fills the chip unevenly
    // (tools/probes/small_sweep.py over 8 .. 520 crops, switch x3_l4_narrow_nmt = 0 / 1000):
    //  * 33 .. 62 crops: four ways split, 272 .. 496 blocks for 256 CUs - the CUs that get two blocks set the time (78 us per convolution
    //    for every size in the range, 58 at 32 crops); 64-wide tiles (three blocks per CU, two ways split) are 12 .. 75 us per pass faster.
    //    Not at 63 / 64 crops (32 tile rows): passes of 64 keep the form the config-1 rank vectors were checked with.
    //  * unsplit launches whose last layer of 256 blocks covers at most half the CUs (129 .. 192, 257 .. 320, 385 .. 448 crops): the
    //    narrow tile halves that tail: -100 .. -160 us per pass (2.32 -> 2.20 ms at 129 crops); past half a layer the wide tile wins by
    //    as much.  Same K order per output element: bit-identical to the wide launch.
    //  * two ways split (65 .. 128 crops): no difference, stays wide.
    if (p.W == 8 && p.N == 512 && ctx->x3_l4_narrow_nmt > 0) {
        const int nmt = (p.M + 255) / 256, ncr = p.Cin / 3 / 32;
        const int t128 = nmt * 4, s128 = x3_split(ctx, t128, ncr, true);
        if (s128 == 4 && t128 * s128 > 256 && nmt <= ctx->x3_l4_narrow_nmt) return false;
        if (s128 == 1 && t128 % 256 != 0 && t128 % 256 <= 128) return false;
        // 65 .. 96 crops: two ways split the 128-wide launch has 264 .. 384 blocks of eight chunks (140 us per convolution at 66 crops against
        // 110 at 64, where 512 blocks of four chunks fill every CU twice); 64-wide tiles two ways split are 528 .. 768 blocks of the same
        // four-chunk size, three to a CU (x3_l4_slots) - same K halves per output element
        if (s128 == 2 && t128 > 128 && t128 * 4 <= 768) return false;
    }
    return true;
}

// block slots the split count may fill: two blocks per CU, and three for layer 4's 64-wide tile at 65 .. 96 crops (x3_wide_tiles).  NOT for its
// 64-wide launches at 33 .. 62 crops: with 768 slots those split four ways (544 .. 768 blocks) instead of two and a pass of 33 .. 47 crops
// took 100 us longer (the pass-size sweep caught it: Poisson(30) mean 772 -> 808 us).
static int x3_slots(const Gemm16Params& p, bool wide) { return (!wide && p.W == 8 && p.N == 512 && (p.M + 255) / 256 > 32) ? 768 : 512; }

template <int TW, int IMGS>
int launch_x3m16(reid_ctx* ctx, const Gemm16Params& p0) {
    Gemm16Params p = p0;
    const int nmt = (p.M + 255) / 256;
    const bool wide = x3_wide_tiles(ctx, p);
    const int tiles = nmt * (wide ? p.N / 128 : p.N / 64);
    const int sk = x3_split(ctx, tiles, p.Cin / 3 / 32, wide, x3_slots(p, wide));
    if (sk > 1) {
        float* ws;
        int* cnt;
        const bool fresh = ctx->ws.find("x3.splitk_cnt") == ctx->ws.end();
        ARG_CHECK(tiles <= 512);      // [tile] arrivals, [512 + tile] readers done: a split launch has at most 768 / 2 tiles (x3_slots)
        REID_TRY(ctx_ws(ctx, "x3.splitk_ws", (size_t)tiles * sk * 256 * (wide ? 128 : 64) * sizeof(float), (void**)&ws));
        REID_TRY(ctx_ws(ctx, "x3.splitk_cnt", 1024 * sizeof(int), (void**)&cnt));
        if (fresh) HIP_TRY(hipMemsetAsync(cnt, 0, 1024 * sizeof(int), ctx->stream));
        p.split_k = sk; p.splitk_ws = ws; p.splitk_cnt = cnt;
    } else {
        p.split_k = 1;
    }
    // the unrolled form reaches the input through a 32-bit buffer descriptor with 0x7fffff00 as "outside"
    const bool unrolled = ctx->x3_unroll && (long long)p.M * (p.Cin / 3) * 4 < 0x7f000000ll && (long long)p.N * p.ldb * 2 < 0x7f000000ll;
    if (wide && unrolled) hipLaunchKernelGGL((conv3x3_x3u_kernel<TW, IMGS, 128, 2>), dim3(tiles * sk), dim3(256), 0, ctx->stream, p);
    else if (wide) hipLaunchKernelGGL((conv3x3_x3m16_kernel<TW, IMGS, 128>), dim3(tiles * sk), dim3(256), 0, ctx->stream, p);
    else if (unrolled && ctx->x3_unroll == 3) hipLaunchKernelGGL((conv3x3_x3u_kernel<TW, IMGS, 64, 3>), dim3(tiles * sk), dim3(256), 0, ctx->stream, p);
#ifdef REID_EXPERIMENTS   // four blocks per CU (spills: 14.05 against 12.32 ms per pass) and groups of three steps per barrier: experiment builds only
    else if (unrolled && ctx->x3_unroll == 4) hipLaunchKernelGGL((conv3x3_x3u_kernel<TW, IMGS, 64, 4>), dim3(tiles * sk), dim3(256), 0, ctx->stream, p);
    else if (unrolled && ctx->x3_unroll == 5) hipLaunchKernelGGL((conv3x3_x3u_kernel<TW, IMGS, 64, 3, 3>), dim3(tiles * sk), dim3(256), 0, ctx->stream, p);
#endif
    else hipLaunchKernelGGL((conv3x3_x3m16_kernel<TW, IMGS, 64>), dim3(tiles * sk), dim3(256), 0, ctx->stream, p);
    return REID_OK;
}

template <int TW, int IMGS>
void launch_x3(reid_ctx* ctx, const Gemm16Params& p) {
    const int nmt = (p.M + 255) / 256;
    if (p.N % 128 == 0) hipLaunchKernelGGL((conv3x3_x3_kernel<TW, IMGS, 128>), dim3(nmt * (p.N / 128)), dim3(256), 0, ctx->stream, p);
    else hipLaunchKernelGGL((conv3x3_x3_kernel<TW, IMGS, 64>), dim3(nmt * (p.N / 64)), dim3(256), 0, ctx->stream, p);
}

}  // namespace

// output tiles of a launch in the width launch_x3m16 picks for it
static int launch_tiles(const reid_ctx* ctx, const Gemm16Params& p) {
    const int nmt = (p.M + 255) / 256;
    return nmt * (x3_wide_tiles(ctx, p) ? p.N / 128 : p.N / 64);
}

// Launches of at least two blocks for every CU, and (round 6) the smaller ones listed below; the smallest keep conv3x3_f16.hip's forms.
bool conv3x3_x3_supported(const reid_ctx* ctx, const Gemm16Params& p) {
    if (!ctx->split_x3 || p.split_terms != 3 || p.Cin % 96 != 0 || p.N % 64 != 0 || p.M % 128 != 0 || !conv3x3_f16_supported(p)) return false;
    if (p.relu_from % 16 != 0 || (p.pack16 && p.pack_from % 32 != 0)) return false;   // x3m16_tail decides ReLU / the packed store per sixteen-column unit
    // 64-wide tiles exist for the 32-wide maps (layer 1) only, in form 3: four blocks per CU, one halo buffer.  Measured at 1024 crops:
    // 570 us per launch against 729 for the 12-wave kernel (and 666 for a two-blocks-per-CU form with groups of three taps per barrier)
    if (p.N % 128 != 0 && ctx->split_x3 < 3) return false;
    const int nmt = (p.M + 255) / 256;
    const long long blocks = (long long)nmt * (p.N % 128 == 0 ? p.N / 128 : p.N / 64);
    if (blocks >= ctx->split_x3_min_blocks) return true;
    if (ctx->split_x3 < 2 || !ctx->split_x3_small) return false;
    if (ctx->split_x3_small == 1) return true;       // every small launch (A/B)
    // Smaller launches (round 6, split_x3_small = 2): since x3m16_tail reduces split-K partials as a reduce-scatter these kernels beat
    // conv3x3_f16.hip's 12-wave blocks wherever a launch is not tiny - measured per layer at 30 / 64 / 120 / 200 crops
    // (gpurun_out/r6/timeline_*): layer 4 at every size (30 crops: 45.5 / 69 / 70 / 73 -> 40 / 59 / 68 / 61 us, 200 crops: 385 -> 290 us),
    // layer 3 likewise (64 crops: 48 -> 41 us; 30 crops: 31 -> 23 us on 64-wide tiles, x3_wide_tiles), the 16- and 32-wide maps from ~400 tiles on
    // (layer 2 at 120 crops: 65 -> 51 us; at 64 crops 37 against 42 - stays; layer 1 at 30 crops 24 against 31-35 - stays).
    // Second pass over every pass size of a tracking stream (tools/probes/small_sweep.py, timelines at 20 / 28 / 33 crops with and without
    // split_x3_small = 1): (i) layers 3 and 4 at EVERY size - 20 crops: layer 3 29-31 -> 21-22 us per convolution, layer 4 36 / 57 / 57 / 58 ->
    // 33 / 54 / 52 / 54; the whole pass 8-22 crops: -30 .. -85 us (21-22 crops had been slower than 23: 762 against 671 us); (ii) layer 1 as
    // soon as the 12-wave kernel's tiles (256 pixels, one block per CU) no longer fit ONE round of the chip - 33 crops: 264 blocks in two
    // rounds 40-43 us against 28-30 us for the 64-wide form at three blocks per CU; up to 32 crops the 12-wave kernel stays (28 crops: 22-24
    // against 25-27).  Layer 2 is a wash either way (+-2 us per convolution at 20 / 28 / 33 crops) and keeps the 12-wave kernel below ~400 tiles.
    const int tiles = launch_tiles(ctx, p);
    if (p.W == 8) return true;
    if (p.W == 32) return p.M / 256 > 256 || tiles >= 384;
    // layer 2 (16-wide maps): the 12-wave kernel splits K two ways up to 128 tiles (64 crops: 39 us against 42 here); past that it runs
    // unsplit on half the chip - 66 / 80 crops: 55-59 us per convolution against 43-44 here (96 crops and up were here already)
    return p.M / 256 > 128 || tiles >= 384;
}

int launch_conv3x3_x3(reid_ctx* ctx, const Gemm16Params& p0) {
    Gemm16Params p = p0;
    p.fault = ctx->fault;
    p.ablate |= ctx->x3_ablate;
    if (ctx->split_x3 >= 2) {                    // the 16x16x32 form (default)
        if (p.W == 32) REID_TRY((launch_x3m16<32, 1>(ctx, p)));
        else if (p.W == 16) REID_TRY((launch_x3m16<16, 1>(ctx, p)));
        else REID_TRY((launch_x3m16<8, 2>(ctx, p)));
    } else if (p.W == 32) launch_x3<32, 1>(ctx, p);
    else if (p.W == 16) launch_x3<16, 1>(ctx, p);
    else launch_x3<8, 2>(ctx, p);
    LAUNCH_CHECK();
    return REID_OK;
}

#ifdef REID_EXPERIMENTS
// ---- chain kernel launcher: W = width of the layer's maps (8: layers 3-4 as image pairs, 128-wide tiles)
int launch_chain(reid_ctx* ctx, const ChainParams& cp, int W) {
    ARG_CHECK(cp.n_img >= 1 && cp.n_img <= 64 && cp.n_stages >= 1 && cp.n_stages <= 8 && cp.counters && W == 8);
    const int blocks = cp.total_items < 512 ? cp.total_items : 512;          // two per CU
    hipLaunchKernelGGL((chain_kernel<8, 2, 128>), dim3(blocks), dim3(256), 0, ctx->stream, cp);
    LAUNCH_CHECK();
    return REID_OK;
}
#endif

// ---- strided 3x3 / 1x1 convolutions (conv_x3s_kernel)
bool conv_x3s_supported(const reid_ctx* ctx, const Gemm16Params& p) {
    if (!ctx->conv_x3s || ctx->split_x3 < 2 || p.split_terms != 3 || p.Cin % 96 != 0 || p.N % 128 != 0 || p.M % 128 != 0) return false;
    if (!((p.R == 1 || p.R == 3) && p.S == p.R && p.pad == (p.R - 1) / 2 && (p.stride == 1 || p.stride == 2))) return false;
    if (p.R == 3 && p.stride == 1) return false;                       // the halo kernels' case
    if (p.relu_from % 16 != 0 || (p.pack16 && p.pack_from % 32 != 0)) return false;
    const long long a_bytes = (long long)(p.M / (p.Ho * p.Wo)) * p.H * p.W * (p.Cin / 3) * 4;
    return a_bytes < 0x7f000000ll && 128ll * p.ldb * 2 < 0x7f000000ll && 256ll * p.ldc * 4 < 0x7f000000ll && (long long)p.M * 2 < 0x7f000000ll;
}

int launch_conv_x3s(reid_ctx* ctx, const Gemm16Params& p0, int kind, double flops, double bytes) {
    Gemm16Params p = p0;
    p.fault = ctx->fault;
    p.ablate = 0;
    const int tiles = ((p.M + 255) / 256) * (p.N / 128), nq = p.R * p.S * (p.Cin / 3 / 32);
    int sk = 1;                      // (tap, chunk) steps split over up to x3s_sk_cap blocks per tile while the launch fits the 512 block slots
    while (ctx->f16_split_k && tiles * sk * 2 <= 512 && nq / (sk * 2) >= 2 && sk * 2 <= ctx->x3s_sk_cap) sk *= 2;
    if (sk > 1) {
        float* ws;
        int* cnt;
        const bool fresh = ctx->ws.find("x3.splitk_cnt") == ctx->ws.end();
        ARG_CHECK(tiles <= 512);
        REID_TRY(ctx_ws(ctx, "x3.splitk_ws", (size_t)tiles * sk * 256 * 128 * sizeof(float), (void**)&ws));
        REID_TRY(ctx_ws(ctx, "x3.splitk_cnt", 1024 * sizeof(int), (void**)&cnt));
        if (fresh) HIP_TRY(hipMemsetAsync(cnt, 0, 1024 * sizeof(int), ctx->stream));
        p.split_k = sk; p.splitk_ws = ws; p.splitk_cnt = cnt;
    } else {
        p.split_k = 1;
    }
    prof_begin(ctx, kind, flops, bytes);
    hipLaunchKernelGGL((conv_x3s_kernel<128>), dim3(tiles * sk), dim3(256), 0, ctx->stream, p);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

// ---- dense form (Swin linears, fp32-class mode): launched for EVERY batch size of a layer it supports - which arithmetic a layer runs
// in must not depend on how many images a pass holds
bool lin_x3_supported(const reid_ctx* ctx, const Gemm16Params& p) {
    const int Kr = p.K / 3;
    return ctx->lin_x3 && ctx->split_x3 >= 2 && p.lin && p.split_terms == 3 && p.K % 96 == 0 && p.N % 128 == 0 && p.n_real == p.N && p.lda == 2 * Kr &&
           p.ldb == 3 * Kr && !p.scat_h && !p.par4 && (p.C32 != nullptr) != (p.pack_out != 0) &&
           256ll * p.lda * 2 < 0x7f000000ll && 128ll * p.ldb * 2 < 0x7f000000ll && 256ll * p.ldc * 4 < 0x7f000000ll;
}

int launch_lin_x3(reid_ctx* ctx, const Gemm16Params& p0, int kind, double flops, double bytes) {
    Gemm16Params p = p0;
    p.fault = ctx->fault;
    p.ablate = ctx->x3_ablate;
    p.stats = nullptr; p.col_scale = nullptr; p.relu = 0; p.split_k = 1;
    if (p.pack_out) {            // [yh | yl'] f16 [M][2 N]
        p.pack16 = p.C;
        p.pack_from = 0;
        p.C32 = nullptr;
    } else {
        p.pack16 = nullptr;
    }
    prof_begin(ctx, kind, flops, bytes);
    hipLaunchKernelGGL((lin_x3_kernel<128>), dim3(((p.M + 255) / 256) * (p.N / 128)), dim3(256), 0, ctx->stream, p);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
