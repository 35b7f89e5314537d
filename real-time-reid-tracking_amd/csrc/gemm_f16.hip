// C[M][N] (f16) = A'[M][K] (f16) . B[N][K]^T (f16), fp32 accumulate, on v_mfma_f32_32x32x16_f16.
// The fast path of the convolution stack (reid_ctx_set_precision(ctx, 1)): activations and weights are stored as
// fp16, every sum is fp32, BN / residual / ReLU / statistics run in fp32 in the epilogue.
//
// Staging is LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, the im2col gather is the per-lane SOURCE
// address, zero padding is a lane pointing at a zero page.  The LDS image is lane-linear ([row][BK*2 bytes]), so the
// bank-conflict fix is an XOR swizzle applied to the source chunk and to the read address: with RPB = rows per 256-B
// bank row and CH = 16-B chunks per row, chunk c of row r sits at position c ^ ((r / RPB) % CH); the 16 rows of a
// ds_read_b128 lane group then cover the 16 slots of the bank row (SQ_LDS_BANK_CONFLICT = 0 measured).
//
// 256 x BN block tile, 8 waves, NST-deep LDS ring.  A loop that keeps ONE K-tile of DMA in flight (issue -> MFMAs ->
// vmcnt(0) -> barrier) is latency-bound (an LDS-DMA takes ~1 us to land under load).  Here NST-1 K-tiles are in
// flight: per iteration a wave waits for its OWN pieces of tile t with a counted s_waitcnt vmcnt((NST-2)*G) (G = DMA
// instructions per wave per tile), passes ONE raw s_barrier (every wave's pieces of tile t have landed, and every wave
// has finished reading tile t-1), refills the slot of tile t-1 with tile t+NST-1, then runs the MFMAs of tile t.
// No vmcnt(0) inside the loop (a __syncthreads() would drain the ring).
//
// Measured model (tools/bench_conv_f16.py): throughput = DMA byte rate x tile intensity BM*BN/(BM+BN) flop/B.  The DMA
// byte rate is ~7.5 TB/s chip-wide with 64-B rows (BK = 32) and ~12+ TB/s with full 128-B lines (BK = 64), independent
// of the inner-loop order, s_setprio, K ordering (tap-major vs channel-chunk-major) - so: BK = 64 and the widest tile
// that keeps the ring in 160 KB of LDS.
// Requirements: M % 128 == 0 (a ragged last 256-row tile is predicated), N % 64 == 0, K % BK == 0.
#include "reid_internal.h"
#include <type_traits>

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#include "lin_math.h"

namespace {

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(uintptr_t)(p))
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

// LIN: the linear-layer epilogue (Swin) instead of the convolution epilogue - a template parameter, not a run-time branch:
// carrying both epilogues cost the 256-wide conv instantiations 66 more spilled VGPRs (72 -> 214 us per launch)
// (launch bound = waves per SIMD: the linear builds up to 128 columns must stay within 128 VGPRs so that TWO blocks share a CU)
// SPLIT: the "fp32-class" form (Gemm16Params: acc_scale / split_terms; conv3x3_f16.hip has the 3x3 stride-1 sibling): im2col over
// split_terms x C virtual channels of a [xh | xl'] tensor, fp32 epilogue (BN, fp32 residual, ReLU from a column on, column sums)
template <int AMODE, int BN, int BK, int NST, int STAG, bool LIN = false, bool SPLIT = false>
__global__ __launch_bounds__(512, (LIN && BN <= 128) ? 4 : 2) void gemm_f16_kernel(const Gemm16Params p_in) {
    constexpr int BM = 256;
    constexpr int WM = BN == 256 ? 2 : 4;         // waves along M
    constexpr int WN = 8 / WM;                    // waves along N
    constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int ROWB = BK * 2;                  // bytes per LDS row
    constexpr int CH = ROWB / 16;                 // 16-B chunks per row
    constexpr int RPI = 1024 / ROWB;              // rows per DMA wave-instruction
    constexpr int RPB = 256 / ROWB;               // rows per 256-B bank row
    constexpr int KS = BK / 16;                   // MFMA k-steps per tile
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int AJ = BM / RPI / 8;              // A DMA instructions per wave per tile
    constexpr int B_INST = BN / RPI;
    constexpr int BJ = B_INST >= 8 ? B_INST / 8 : 1;
    constexpr int G = AJ + BJ;
    static_assert(NST * STAGE <= 160 * 1024, "LDS ring too large");
    __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    Gemm16Params p = p_in;
    const int nnt = p.N / BN;
    const int nwg = gridDim.x;
    int mtile, ntile;
    {   // XCD-aware, bijective block remap (blocks b and b+8 share an XCD)
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        if constexpr (LIN && AMODE == A16_IM2COL) {
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

    // ---- DMA descriptors: one wave-instruction fills RPI rows; wave w issues A instructions w*AJ + j
    int a_chunk[AJ], a_img[AJ], a_iy0[AJ], a_ix0[AJ];
    bool a_ok[AJ];
    long long a_base[AJ], a_pixc[AJ];    // a_pixc: A16_IM2COL, element offset of the row's tap-(0,0) input pixel (may point before it: pad)
    const int im_acin = (p.split_terms > 0) ? p.Cin / p.split_terms * 2 : p.Cin;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = (wave * AJ + j) * RPI + lane / CH;
        a_chunk[j] = (lane % CH) ^ ((row / RPB) % CH);
        int m = m_blk + row;
        a_ok[j] = m < p.M;
        if (!a_ok[j]) m = 0;  // rows past M (ragged last tile) read the zero page / row 0 and are never stored
        if constexpr (AMODE == A16_DENSE) {
            a_base[j] = (long long)m * p.lda;
        } else {
            const int hw = p.Ho * p.Wo;
            const int img = m / hw, rem = m - img * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_iy0[j] = oy * p.stride - ((LIN && p.asym) ? p.pad_y : p.pad);
            a_ix0[j] = ox * p.stride - ((LIN && p.asym) ? p.pad_x : p.pad);
            a_img[j] = img;
            a_pixc[j] = (((long long)img * p.H + a_iy0[j]) * p.W + a_ix0[j]) * im_acin;
        }
    }
    int b_chunk[BJ], b_row[BJ];
    long long b_base[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        // fewer than 8 B instructions (BN = 64, BK = 32): waves 4-7 repeat waves 0-3 (same bytes to the same place) so
        // that every wave issues the same number of DMAs and the counted waits stay uniform
        const int inst = B_INST >= 8 ? wave * BJ + j : wave % B_INST;
        b_row[j] = inst * RPI;
        const int row = inst * RPI + lane / CH;
        b_chunk[j] = (lane % CH) ^ ((row / RPB) % CH);
        b_base[j] = (long long)(n_blk + row) * p.ldb;
    }

    int im_r = 0, im_s = 0, im_c0 = 0;     // A16_IM2COL: kernel row / column / first channel of the next K-tile to be staged
    // one DMA piece (wave-instruction) of K-tile kt: q < AJ -> A piece q, else B piece q - AJ
    auto stage_piece = [&](int kt, int slot, int q) {
        char* As = lds + slot * STAGE;
        char* Bs = As + A_BYTES;
        const int k0 = kt * BK;
        if (q >= AJ) {
            const int j = q - AJ;
            __builtin_amdgcn_global_load_lds(GPTR(p.B + b_base[j] + k0 + b_chunk[j] * 8), LPTR(Bs + b_row[j] * ROWB), 16, 0, 0);
            return;
        }
        const int j = q;
        if constexpr (AMODE == A16_DENSE) {
            int ka = k0;
            if constexpr (LIN) {   // fp32-class linears: K = 3 Kr virtual columns over A = [xh | xl'] (the last third re-reads xh)
                if (p.split_terms) ka = k0 % p.a_k;
            }
            __builtin_amdgcn_global_load_lds(GPTR(p.A + a_base[j] + ka + a_chunk[j] * 8), LPTR(As + (wave * AJ + j) * 1024), 16, 0, 0);
        } else if constexpr (AMODE == A16_IM2COL) {
            // K order (tap, channel); Cin % BK == 0.  (A channel-chunk-major order that lets the nine taps re-read the same
            // lines back to back was measured: no change.)  The K-tiles are staged in order, so (kernel row, kernel column,
            // first channel) of tile kt are counters that stage() advances - k0 / Cin, tap / S and the fold's modulo used to be
            // three run-time integer divisions per DMA piece, ~300 vector instructions per wave and K-tile beside 16 MFMAs: the
            // Swin trunk's convolutions (K loops of 12-48 tiles) spent their time there.
            int c0 = im_c0;
            int a_cin = p.Cin;
            if constexpr (SPLIT || LIN) {   // p.Cin virtual channels over a tensor of 2C: [xh | xl' | xh (| xl')]
                if (SPLIT || p.split_terms) {
                    a_cin = im_acin;
                    if (c0 >= a_cin) c0 -= a_cin;          // Cin <= 2 a_cin
                }
            }
            const int r = im_r, s = im_s;
            const int iy = a_iy0[j] + r, ix = a_ix0[j] + s;
            const bool ok = a_ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            // element offset = (pixel of tap (0,0), made once per lane) + (tap offset and channel: wave-uniform) - the 64-bit
            // multiplies of ((img H + iy) W + ix) a_cin per DMA piece were quarter-rate vector instructions, ~500 cycles per wave
            // and K-tile beside 128-512 of MFMA
            const long long uni = (long long)(r * p.W + s) * a_cin + c0;
            const f16* src = ok ? p.A + a_pixc[j] + uni + a_chunk[j] * 8 : p.zero_page;
            __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(As + (wave * AJ + j) * 1024), 16, 0, 0);
        } else {  // A16_STEM: zero-padded NHWC4 image, k = r*32 + s*4 + c; a 16-B chunk = one pixel pair of one kernel row
            const int kq = kt * CH + a_chunk[j];   // chunk index along K: kernel row = kq / 4, pixel pair = kq % 4
            const long long pix = ((long long)a_img[j] * p.Hp + (a_iy0[j] + 3 + (kq >> 2))) * p.Wp + (a_ix0[j] + 3 + 2 * (kq & 3));
            __builtin_amdgcn_global_load_lds(GPTR(p.A + pix * 4), LPTR(As + (wave * AJ + j) * 1024), 16, 0, 0);
        }
    };
    auto stage = [&](int kt, int slot) {
#pragma unroll
        for (int q = 0; q < G; ++q) stage_piece(kt, slot, q);
        if constexpr (AMODE == A16_IM2COL) {      // the next tile's position in the (kernel row, kernel column, channel) walk
            im_c0 += BK;
            if (im_c0 == p.Cin) {
                im_c0 = 0;
                if (++im_s == p.S) { im_s = 0; ++im_r; }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int nk = p.K / BK;
    const int swz = (li / RPB) % CH;
    const int a_row_off = (wm * WTM + li) * ROWB;
    const int b_row_off = (wn * WTN + li) * ROWB;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) stage(s, s);

    int slot_c = 0, slot_i = NST - 1;  // ring slots of the tile being computed / refilled
    if constexpr (STAG == 0 || STAG == 2) {
        // STAG == 2: diagnostic build of the same loop with s_memtime stamps (never used by the product path):
        // cycles spent in [vmcnt wait] [barrier] [DMA issue] [fragment reads + MFMAs] summed over the K loop
        unsigned long long t_wait = 0, t_bar = 0, t_issue = 0, t_comp = 0, ta = 0, tb = 0;
#define STAMP(v) if constexpr (STAG == 2) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
        for (int kt = 0; kt < nk; ++kt) {
            const int rem = nk - 1 - kt;
            STAMP(ta);
            if (NST > 2 && rem >= NST - 2) WAIT_VMCNT((NST - 2) * G);
            else if (NST > 3 && rem == 1) WAIT_VMCNT(G);
            else WAIT_VMCNT(0);
            STAMP(tb);
            t_wait += tb - ta;
            RAW_BARRIER();
            STAMP(ta);
            t_bar += ta - tb;
            // (Issuing the refill in pieces between the MFMA clusters of the k-steps instead was measured: 5-15 % slower.)
            if (kt + NST - 1 < nk) stage(kt + NST - 1, slot_i);
            STAMP(tb);
            t_issue += tb - ta;
            const char* As = lds + slot_c * STAGE;
            const char* Bs = As + A_BYTES;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int pos = ((kk * 2 + lh) ^ swz) * 16;
                half8 af[TM], bf[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) af[a] = *(const half8*)(As + a_row_off + a * 32 * ROWB + pos);
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[b] = *(const half8*)(Bs + b_row_off + b * 32 * ROWB + pos);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
            STAMP(ta);
            t_comp += ta - tb;
            slot_c = slot_c + 1 == NST ? 0 : slot_c + 1;
            slot_i = slot_i + 1 == NST ? 0 : slot_i + 1;
        }
        if constexpr (STAG == 2) {
            if (p.diag && lane == 0 && blockIdx.x < 64) {
                unsigned long long* d = p.diag + ((long long)blockIdx.x * 8 + wave) * 4;
                d[0] = t_wait; d[1] = t_bar; d[2] = t_issue; d[3] = t_comp;
            }
        }
#undef STAMP
    } else {
        // Staggered schedule.  Waves w and w+4 share a SIMD; all eight running the same phase at the same time leaves
        // the matrix pipe idle whenever they are all reading LDS / issuing DMA (measured: ~1 PF ceiling even on a dense
        // 8192^3 GEMM).  Each K-tile is split into a LOAD half (issue the ring refill, read every fragment of the tile
        // into registers) and an MFMA half (registers only); waves 4-7 run half a tile behind waves 0-3, with a block
        // barrier between half-slots, so one group's LOAD half always sits beside the other group's MFMA half:
        //     slot 2t   : A = LOAD(t)            B = MFMA(t-1)
        //     slot 2t+1 : A = MFMA(t) + wait     B = LOAD(t) + wait
        // "wait" = this wave's own DMA pieces of tile t+1, counted; the barrier after it publishes the tile to all.
        static_assert(NST >= 3, "the staggered schedule needs at least three ring slots");
        const bool grpA = wave < 4;
        half8 af[KS][TM], bf[KS][TN];
        auto load_half = [&](int kt) {
            if (kt + NST - 1 < nk) stage(kt + NST - 1, slot_i);
            const char* As = lds + slot_c * STAGE;
            const char* Bs = As + A_BYTES;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int pos = ((kk * 2 + lh) ^ swz) * 16;
#pragma unroll
                for (int a = 0; a < TM; ++a) af[kk][a] = *(const half8*)(As + a_row_off + a * 32 * ROWB + pos);
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[kk][b] = *(const half8*)(Bs + b_row_off + b * 32 * ROWB + pos);
            }
            slot_c = slot_c + 1 == NST ? 0 : slot_c + 1;
            slot_i = slot_i + 1 == NST ? 0 : slot_i + 1;
        };
        auto mfma_half = [&]() {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk][a], bf[kk][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        auto wait_next = [&](int kt) {   // own pieces of tile kt+1 (tiles kt+2 .. may stay in flight)
            const int rem = nk - 2 - kt;  // tiles issued after kt+1
            if (rem >= NST - 2) WAIT_VMCNT((NST - 2) * G);
            else if (NST > 3 && rem == 1) WAIT_VMCNT(G);
            else WAIT_VMCNT(0);
        };
        // tile 0 landed for everyone
        if (nk - 1 >= NST - 2) WAIT_VMCNT((NST - 2) * G);
        else if (NST > 3 && nk - 1 == 1) WAIT_VMCNT(G);
        else WAIT_VMCNT(0);
        RAW_BARRIER();
        for (int kt = 0; kt <= nk; ++kt) {
            if (grpA) {
                if (kt < nk) load_half(kt);
            } else {
                if (kt > 0) mfma_half();
            }
            RAW_BARRIER();
            if (kt < nk) {
                if (grpA) mfma_half();
                else load_half(kt);
                wait_next(kt);
            }
            RAW_BARRIER();
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ epilogue (fp32 math, f16 stores)
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int ldc = (int)p.ldc;
    f16* Cb = p.C + (long long)m_blk * ldc + n_blk;
    const f16* Rb = p.residual ? p.residual + (long long)m_blk * ldc + n_blk : nullptr;
    const int row0 = wm * WTM + 4 * lh;
    const int m_valid = p.M - m_blk;  // rows of this tile that exist (>= 256 except for a ragged last tile)

    if constexpr (SPLIT) {   // fp32 epilogue straight from the accumulators (see conv3x3_f16.hip, SPLIT build)
        float s1[TN], s2[TN];
        float vmax = 0.f;   // largest packed magnitude (range guard, conv3x3_f16.hip)
        // lean form, as in conv3x3_f16.hip: buffer instructions with the row in the scalar offset, whole waves skipped in a ragged
        // tile (M % 128 == 0: its 128 rows belong to the waves wm 0, 1); the arithmetic is the general loop's
        static_assert(WM == 4, "the SPLIT build is the 128-wide tile: four waves along M, 64 rows each");
        const bool lean = !p.general_epi && (long long)BM * ldc * 4 < 0x7fffff00ll && (long long)BM * 2 * p.N * 2 < 0x7fffff00ll &&
                          (m_valid >= BM || m_valid == 128);
        if (lean) {
            const bool wave_live = m_valid >= BM || wm < 2;
            const int col0 = n_blk + wn * WTN + li;
            const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C32 + (long long)m_blk * ldc), 0, BM * ldc * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rs =
                __builtin_amdgcn_make_buffer_rsrc((void*)((p.res32 ? p.res32 : p.C32) + (long long)m_blk * ldc), 0, BM * ldc * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t k_rs =
                __builtin_amdgcn_make_buffer_rsrc((void*)((p.pack16 ? p.pack16 : (f16*)p.C32) + (long long)m_blk * 2 * p.N), 0, BM * 2 * p.N * 2, 0x00020000);
            const int voff = (4 * lh * ldc + col0) * 4, koff = (4 * lh * 2 * p.N + col0) * 2;
            auto run = [&](auto res_c) {
                constexpr bool RES = decltype(res_c)::value;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int col = col0 + b * 32;
                    const float cs = (p.col_scale ? p.col_scale[col] : 1.f) * p.acc_scale;
                    const float sh = p.col_scale ? p.col_shift[col] : 0.f;
                    const float lo = (p.relu && col >= p.relu_from) ? 0.f : -INFINITY;
                    const bool pk = p.pack16 && col >= p.pack_from;
                    float t1 = 0.f, t2 = 0.f;
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        float r[16];
                        if constexpr (RES) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int urow = wm * WTM + a * 32 + (e & 3) + 8 * (e >> 2);
                                r[e] = wave_live ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, voff + b * 128, urow * ldc * 4, 0)) : 0.f;
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int urow = wm * WTM + a * 32 + (e & 3) + 8 * (e >> 2);
                            float v = acc[a][b][e] * cs + sh;
                            v += RES ? r[e] : 0.f;
                            v = fmaxf(v, lo);
                            if (wave_live) {
                                t1 += v;
                                t2 += v * v;
                                if (pk) {
                                    vmax = fmaxf(vmax, fabsf(v));
                                    const f16 hv = (f16)v;
                                    const f16 lv = (f16)((v - (float)hv) * 2048.0f);
                                    // (pairing neighbouring lanes' values into one dword store per lane - DPP quad_perm - was measured: slower)
                                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), k_rs, koff + b * 64, urow * 2 * p.N * 2, 0);
                                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, lv), k_rs, koff + b * 64 + p.N * 2, urow * 2 * p.N * 2, 0);
                                } else {
                                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rs, voff + b * 128, urow * ldc * 4, 0);
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
            const int lcol = wn * WTN + b * 32 + li;
            const int col = n_blk + lcol;
            const float cs = (p.col_scale ? p.col_scale[col] : 1.f) * p.acc_scale;
            const float sh = p.col_scale ? p.col_shift[col] : 0.f;
            const float lo = (p.relu && col >= p.relu_from) ? 0.f : -INFINITY;
            const bool pk = p.pack16 && col >= p.pack_from;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                float r[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = row0 + a * 32 + (e & 3) + 8 * (e >> 2);
                    const int rc = rl < m_valid ? rl : 0;
                    r[e] = p.res32 ? p.res32[(long long)(m_blk + rc) * ldc + col] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = row0 + a * 32 + (e & 3) + 8 * (e >> 2);
                    float v = acc[a][b][e] * cs + sh;
                    v += r[e];
                    v = fmaxf(v, lo);
                    if (rl < m_valid) {
                        t1 += v;
                        t2 += v * v;
                        if (pk) {   // [yh | yl'] for the next convolution's loader (uniform per 32-column tile: pack_from % 32 == 0)
                            vmax = fmaxf(vmax, fabsf(v));
                            const f16 hv = (f16)v;
                            f16* dst = p.pack16 + (long long)(m_blk + rl) * 2 * p.N + col;
                            dst[0] = hv;
                            dst[p.N] = (f16)((v - (float)hv) * 2048.0f);
                        } else {
                            p.C32[(long long)(m_blk + rl) * ldc + col] = v;
                        }
                    }
                }
            }
            s1[b] = t1;
            s2[b] = t2;
        }
        }
        if (p.fault && !(vmax < 65504.f)) p.fault[0] = 1;
        if (p.stats) {   // per 128-row tile: the block covers two of them
            float* stat_lds = (float*)lds;  // [WM][BN][2]
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int lcol = wn * WTN + b * 32 + li;
                const float t1 = s1[b] + __shfl_xor(s1[b], 32);
                const float t2 = s2[b] + __shfl_xor(s2[b], 32);
                if (lh == 0) {
                    stat_lds[(wm * BN + lcol) * 2 + 0] = t1;
                    stat_lds[(wm * BN + lcol) * 2 + 1] = t2;
                }
            }
            __syncthreads();
            for (int t = tid; t < 2 * BN; t += 512) {
                const int half = t / BN, c = t - half * BN;
                if (half * 128 >= m_valid) continue;
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < WM / 2; ++w) {
                    t1 += stat_lds[((half * (WM / 2) + w) * BN + c) * 2 + 0];
                    t2 += stat_lds[((half * (WM / 2) + w) * BN + c) * 2 + 1];
                }
                float* o = p.stats + ((long long)(mtile * 2 + half) * p.N + n_blk + c) * 2;
                o[0] = t1;
                o[1] = t2;
            }
        }
        return;
    }
    if constexpr (LIN) {   // linear layers (Swin): bias, erf-GELU, fp32 residual stream, fp32 or f16 output, ragged M and N
#if defined(__HIP_DEVICE_COMPILE__)
        // Swin's K loops are 3-24 tiles long: the epilogue decides these launches.  Its two hot forms carry no per-element
        // address arithmetic, predicate or run-time branch; everything else takes the general loop at the end.
        const int n_real = p.n_real ? p.n_real : p.N;
        const float asc = p.split_terms ? p.acc_scale : 1.0f;   // fp32-class linears: the accumulator holds 2^11 x the product
        constexpr bool LIN_LDS = BM * BN * 2 <= NST * STAGE;
        // (1) f16 outputs that are plain row-major (qkv, fc1): staged through LDS, leave as whole 16-byte pieces of a row
        const bool staged = LIN_LDS && !p.C32 && !p.res32 && p.scat_h == 0 && (n_real & 7) == 0;
        if (staged) {
            f16* tile = (f16*)lds;
            auto fill = [&](auto act_c) {
                constexpr bool ACT = decltype(act_c)::value;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int lcol = wn * WTN + b * 32 + li;
                    const int col = n_blk + lcol;
                    const float bias = (p.col_shift && col < n_real) ? p.col_shift[col] : 0.f;
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            float v = fmaf(acc[a][b][e], asc, bias);
                            if constexpr (ACT) v = gelu_f16_storage(v);
                            acc[a][b][e] = v;   // kept for the low tile of a packed output
                            tile[(row0 + a * 32 + (e & 3) + 8 * (e >> 2)) * BN + lcol] = cvt_f16_rn(v);
                        }
                }
            };
            if (p.act == 1) fill(std::true_type{});
            else fill(std::false_type{});
            __syncthreads();
            constexpr int C8 = BN / 8;
            for (int idx = tid; idx < BM * C8; idx += 512) {
                const int row = idx / C8, c8 = idx - row * C8;
                if (row < m_valid && n_blk + c8 * 8 < n_real)
                    *(half8*)(p.C + (long long)(m_blk + row) * ldc + n_blk + c8 * 8) = *(const half8*)(tile + row * BN + c8 * 8);
            }
            if (p.pack_out) {   // [yh | yl']: the low tile (yl' = f16((y - yh) 2^11)) goes n_real columns further
                __syncthreads();
                float vmax = 0.f;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int lcol = wn * WTN + b * 32 + li;
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float v = acc[a][b][e];
                            vmax = fmaxf(vmax, fabsf(v));
                            tile[(row0 + a * 32 + (e & 3) + 8 * (e >> 2)) * BN + lcol] = cvt_f16_rn((v - (float)cvt_f16_rn(v)) * 2048.0f);
                        }
                }
                if (p.fault && !(vmax < 65504.f)) p.fault[0] = 1;   // range guard (rows past M hold finite zeros-input results)
                __syncthreads();
                for (int idx = tid; idx < BM * C8; idx += 512) {
                    const int row = idx / C8, c8 = idx - row * C8;
                    if (row < m_valid && n_blk + c8 * 8 < n_real)
                        *(half8*)(p.C + (long long)(m_blk + row) * ldc + n_real + n_blk + c8 * 8) = *(const half8*)(tile + row * BN + c8 * 8);
                }
            }
            return;
        }
        // (2) the fp32 residual stream (proj, fc2, patch merging): buffer loads / stores from the MFMA layout, 128 B per row and
        // instruction; per-lane byte offset fixed per column block, row offset in the SGPR operand, columns past N fall outside
        // the descriptor and are dropped by the hardware
        // (the SGPR offset is not part of the range check: a tile with fewer than BM rows takes the general loop)
        if (p.C32 && p.scat_h == 0 && (long long)BM * ldc * 4 < 0x7fffff00ll && m_valid >= BM) {
            const int rows = BM;
            const int recs = (int)(((long long)(rows - 1) * ldc + n_real) * 4);
            const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C32 + (long long)m_blk * ldc), 0, recs, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rs =
                __builtin_amdgcn_make_buffer_rsrc((void*)((p.res32 ? p.res32 : p.C32) + (long long)m_blk * ldc), 0, recs, 0x00020000);
            auto stream = [&](auto act_c, auto res_c) {
                constexpr bool ACT = decltype(act_c)::value, RES = decltype(res_c)::value;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int col = n_blk + wn * WTN + b * 32 + li;
                    const bool col_ok = col < n_real;
                    const float bias = (p.col_shift && col_ok) ? p.col_shift[col] : 0.f;
                    const int voff = col_ok ? (row0 * ldc + col) * 4 : 0x7fffff00;
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
                            float v = fmaf(acc[a][b][e], asc, bias);
                            if constexpr (ACT) v = gelu_f16_storage(v);
                            if constexpr (RES) v += res[e];
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rs, voff, (a * 32 + (e & 3) + 8 * (e >> 2)) * ldc * 4, 0);
                        }
                    }
                }
            };
            if (p.act == 1) {
                if (p.res32) stream(std::true_type{}, std::true_type{});
                else stream(std::true_type{}, std::false_type{});
            } else {
                if (p.res32) stream(std::false_type{}, std::true_type{});
                else stream(std::false_type{}, std::false_type{});
            }
            return;
        }
        // (3) everything else (ConvTranspose parity scatter, f16 outputs with odd widths): element by element, the output row of
        // each of the lane's 32 rows computed once (divisions by multiply-high with host-made reciprocals)
        float bias[TN];
        int colv[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            colv[b] = n_blk + wn * WTN + b * 32 + li;
            bias[b] = (p.col_shift && colv[b] < n_real) ? p.col_shift[colv[b]] : 0.f;
        }
        auto general = [&](auto scat_c) {
            constexpr bool SCAT = decltype(scat_c)::value;
            const unsigned hw = (unsigned)(p.scat_h * p.scat_w);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = row0 + a * 32 + (e & 3) + 8 * (e >> 2);
                    if (rl >= m_valid) continue;
                    const unsigned row = (unsigned)(m_blk + rl);
                    long long orow = row;
                    if constexpr (SCAT) {   // ConvTranspose2d(4,2,1) output parity: (img, j, i) -> (img, 2j+py, 2i+px)
                        const unsigned img = __umulhi(row, p.scat_mhw), rem = row - img * hw;
                        const unsigned jj = __umulhi(rem, p.scat_mw), ii = rem - jj * (unsigned)p.scat_w;
                        orow = ((long long)img * 2 * p.scat_h + 2 * jj + p.scat_py) * (2 * p.scat_w) + 2 * ii + p.scat_px;
                    }
                    const long long obase = orow * ldc;
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        if (colv[b] >= n_real) continue;
                        float v = fmaf(acc[a][b][e], asc, bias[b]);   // the same explicit operations as the two fast forms: a ragged last
                        if (p.act == 1) v = gelu_f16_storage(v);        // tile must give its rows what a full tile would
                        const long long o = obase + colv[b];
                        if (p.res32) v += p.res32[o];
                        if (p.C32) p.C32[o] = v;
                        else p.C[o] = cvt_f16_rn(v);
                    }
                }
        };
        if (p.scat_h > 0) general(std::true_type{});
        else general(std::false_type{});
#endif
        return;
    }

    // The residual tile [256][BN] f16 is brought into the (now free) LDS ring by DMA and read from there: per-lane
    // 2-byte global loads would be 128 latency-serialised round trips per lane (measured +40 us per launch).
    constexpr bool RES_LDS = BM * BN * 2 <= NST * STAGE;
    if (RES_LDS && Rb) {
        constexpr int RROW = BN * 2, RCH = RROW / 16, RRPI = 1024 / RROW, RJ = BM / RRPI / 8;
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int inst = wave * RJ + j;
            const int row = inst * RRPI + lane / RCH;
            const f16* src = row < m_valid ? Rb + (long long)row * ldc + (lane % RCH) * 8 : p.zero_page;
            __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(lds + inst * 1024), 16, 0, 0);
        }
        __syncthreads();  // drains the DMA (vmcnt(0)) and publishes it
    }
    const f16* Rl = (const f16*)lds;

    // pass 1: BN / residual / ReLU in fp32, in place in the accumulators; per-column partial sums.  Straight-line and
    // branch-free: rows past m_valid (ragged last tile) all belong to the waves of the second 128-row half, whose
    // statistics and stores are skipped as a whole, so no per-element predicate is needed; flags become operands.
    float s1[TN], s2[TN];
    const float lo = p.relu ? 0.f : -INFINITY;
    auto pass1 = [&](auto with_res) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int lcol = wn * WTN + b * 32 + li;
            const int col = n_blk + lcol;
            const float cs = p.col_scale ? p.col_scale[col] : 1.f;
            const float sh = p.col_scale ? p.col_shift[col] : 0.f;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                float r[16];
                if constexpr (decltype(with_res)::value) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rl = row0 + a * 32 + (e & 3) + 8 * (e >> 2);
                        if constexpr (RES_LDS) r[e] = (float)Rl[rl * BN + lcol];
                        else r[e] = rl < m_valid ? (float)Rb[rl * ldc + lcol] : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[a][b][e] * cs + sh;
                    if constexpr (decltype(with_res)::value) v += r[e];
                    v = fmaxf(v, lo);
                    t1 += v;
                    t2 += v * v;
                    acc[a][b][e] = v;
                }
            }
            s1[b] = t1;
            s2[b] = t2;
        }
    };
    if (Rb) pass1(std::true_type{});
    else pass1(std::false_type{});
    __syncthreads();   // every wave is done with the residual tile
    if (p.stats) {
        // statistics are kept per 128-row tile (= per image for the 16x8 maps): the block covers two of them
        float* stat_lds = (float*)lds;  // [WM][BN][2]
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int lcol = wn * WTN + b * 32 + li;
            const float t1 = s1[b] + __shfl_xor(s1[b], 32);
            const float t2 = s2[b] + __shfl_xor(s2[b], 32);
            if (lh == 0) {
                stat_lds[(wm * BN + lcol) * 2 + 0] = t1;
                stat_lds[(wm * BN + lcol) * 2 + 1] = t2;
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * BN; t += 512) {
            const int half = t / BN, c = t - half * BN;
            if (half * 128 >= m_valid) continue;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM / 2; ++w) {
                t1 += stat_lds[((half * (WM / 2) + w) * BN + c) * 2 + 0];
                t2 += stat_lds[((half * (WM / 2) + w) * BN + c) * 2 + 1];
            }
            float* o = p.stats + ((long long)(mtile * 2 + half) * p.N + n_blk + c) * 2;
            o[0] = t1;
            o[1] = t2;
        }
        __syncthreads();
    }
    // pass 2: the output tile goes through LDS so that global stores are whole 16-byte pieces of a row (the per-lane
    // 2-byte stores of the MFMA layout cost a third of the block's time: one block per CU, nothing overlaps the tail)
    constexpr bool OUT_LDS = BM * BN * 2 <= NST * STAGE;
    if constexpr (OUT_LDS) {
        f16* tile = (f16*)lds;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int lcol = wn * WTN + b * 32 + li;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) tile[(row0 + a * 32 + (e & 3) + 8 * (e >> 2)) * BN + lcol] = (f16)acc[a][b][e];
        }
        __syncthreads();
        constexpr int C8 = BN / 8;
        for (int idx = tid; idx < BM * C8; idx += 512) {
            const int row = idx / C8, c8 = idx - row * C8;
            if (row < m_valid) *(half8*)(Cb + (long long)row * ldc + c8 * 8) = *(const half8*)(tile + row * BN + c8 * 8);
        }
    } else {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int lcol = wn * WTN + b * 32 + li;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = row0 + a * 32 + (e & 3) + 8 * (e >> 2);
                    if (rl < m_valid) Cb[rl * ldc + lcol] = (f16)acc[a][b][e];
                }
        }
    }
}

template <int AMODE, int BN, int BK, int NST, int STAG = 0, bool LIN = false>
int launch_cfg(reid_ctx* ctx, const Gemm16Params& p) {
    if (p.K % BK != 0 || (AMODE == A16_IM2COL && p.Cin % BK != 0)) {
        reid_set_error("gemm_f16: K=%d / Cin=%d not a multiple of BK=%d", p.K, p.Cin, BK);
        return REID_ERR_ARG;
    }
    hipLaunchKernelGGL((gemm_f16_kernel<AMODE, BN, BK, NST, STAG, LIN>), dim3(((p.M + 255) / 256) * (p.N / BN) * (p.par4 ? 4 : 1)), dim3(512), 0,
                       ctx->stream, p);
    LAUNCH_CHECK();
    return REID_OK;
}

// tile / ring configuration: cfg = BN*1000 + BK*10 + NST (ctx->f16_cfg forces one, 0 = heuristic)
template <int AMODE>
int launch_any(reid_ctx* ctx, const Gemm16Params& p) {
    int cfg = ctx->f16_cfg;
    if (cfg == 0 || p.N % ((cfg % 1000000) / 1000) != 0) {
        // measured per layer shape with tools/bench_conv_f16.py (256 crops): Cout 64 -> 64642 (454 TF), 128 -> 128323
        // (624 TF), 256 -> 128642 (679 TF), 512 -> 256642 (953 TF) when that still gives >= 192 blocks
        const long long mt = (p.M + 255) / 256;
        const bool k64 = p.K % 64 == 0 && (AMODE != A16_IM2COL || p.Cin % 64 == 0);
        // (the linear-epilogue build of the 256-wide tile used to spill 98 VGPRs: the fp16-storage mode's launches take the 128-wide one)
        int bn = !p.lin && p.N % 256 == 0 && mt * (p.N / 256) >= 192 ? 256 : (p.N % 128 == 0 ? 128 : 64);
        // fp32-class Swin layers whose width allows it and whose K loop is long (stage 3 fc1, stage 4, the 768-wide trunk convolutions): the
        // 256-wide tile's main loop is 20-25 % faster on these shapes (profiles/r04_gemm_cfg_sweep.txt) and its linear epilogue no longer
        // spills more than the 128-wide one's (23 VGPRs each).  The choice depends on the layer's shape only, and an output's K order is
        // the same in both tiles: bit-identical results.
        if (p.lin && ctx->f16_lin_256 && p.split_terms && p.N % 256 == 0 && k64 && p.K >= 1152) bn = 256;
        // a tracking frame: twice the blocks on a half-empty chip.  Convolutions only: the tile shape of a Swin linear must not
        // depend on the batch (the two instantiations round their epilogues differently, an image's embedding would too)
        if (!p.lin && bn == 128 && p.N > 128 && mt * (p.N / 128) < 128) bn = 64;
        if (!k64) cfg = bn * 1000 + 320 + (bn == 256 ? 4 : 3);
        else if (bn == 128 && p.N == 128) cfg = 128323;
        // Swin's linears have short K loops (K = 96 .. 768, fc2 up to 3072): with BK = 32 and three stages a block needs 72 KB of
        // LDS, so TWO blocks share a CU and one's prologue / epilogue hides behind the other's MFMAs (measured: 15.97 -> 17.7 k img/s)
        else if (p.lin) cfg = bn == 256 ? 256642 : bn * 1000 + 323;
        else cfg = bn * 1000 + 642;
    }
    if (p.lin) {   // linear-epilogue builds exist for the tile shapes the heuristic above picks (no STEM mode)
        if constexpr (AMODE != A16_STEM) {
            switch (cfg) {
                case 256324: return launch_cfg<AMODE, 256, 32, 4, 0, true>(ctx, p);
                case 256642: return launch_cfg<AMODE, 256, 64, 2, 0, true>(ctx, p);
                case 128323: return launch_cfg<AMODE, 128, 32, 3, 0, true>(ctx, p);
                case 64323: return launch_cfg<AMODE, 64, 32, 3, 0, true>(ctx, p);
#ifdef REID_EXPERIMENTS
                case 64642: return launch_cfg<AMODE, 64, 64, 2, 0, true>(ctx, p);
#endif
                default: break;
            }
        }
        reid_set_error("gemm_f16: no linear-epilogue build of tile configuration %d", cfg);
        return REID_ERR_ARG;
    }
    // The product library carries the builds the heuristic above can pick; every other tile / ring configuration, the staggered
    // schedule (+1000000) and the stamped diagnostic builds (+3000000) - 17 more per A mode, most of them spilling - exist in builds
    // made with -DREID_EXPERIMENTS only (make EXPERIMENTS=1: tools/gemm_cfg_sweep.py, tools/diag_gemm_f16.py).
    if constexpr (AMODE == A16_STEM) {           // the 7x7 stem: 64 output channels
        switch (cfg) {
            case 64323: return launch_cfg<AMODE, 64, 32, 3>(ctx, p);
            case 64642: return launch_cfg<AMODE, 64, 64, 2>(ctx, p);
            default: break;
        }
    } else {
        switch (cfg) {
            case 256324: return launch_cfg<AMODE, 256, 32, 4>(ctx, p);
            case 256642: return launch_cfg<AMODE, 256, 64, 2>(ctx, p);
            case 128323: return launch_cfg<AMODE, 128, 32, 3>(ctx, p);
            case 128642: return launch_cfg<AMODE, 128, 64, 2>(ctx, p);
            case 64323: return launch_cfg<AMODE, 64, 32, 3>(ctx, p);
            case 64642: return launch_cfg<AMODE, 64, 64, 2>(ctx, p);
            default: break;
        }
    }
#ifdef REID_EXPERIMENTS
    switch (cfg) {
        case 256324: return launch_cfg<AMODE, 256, 32, 4>(ctx, p);
        case 256642: return launch_cfg<AMODE, 256, 64, 2>(ctx, p);
        case 128323: return launch_cfg<AMODE, 128, 32, 3>(ctx, p);
        case 128642: return launch_cfg<AMODE, 128, 64, 2>(ctx, p);
        case 256323: return launch_cfg<AMODE, 256, 32, 3>(ctx, p);
        case 128324: return launch_cfg<AMODE, 128, 32, 4>(ctx, p);
        case 128643: return launch_cfg<AMODE, 128, 64, 3>(ctx, p);
        case 64324: return launch_cfg<AMODE, 64, 32, 4>(ctx, p);
        case 64643: return launch_cfg<AMODE, 64, 64, 3>(ctx, p);
        // staggered schedule: +1000000
        case 1256324: return launch_cfg<AMODE, 256, 32, 4, 1>(ctx, p);
        case 1256323: return launch_cfg<AMODE, 256, 32, 3, 1>(ctx, p);
        case 1128324: return launch_cfg<AMODE, 128, 32, 4, 1>(ctx, p);
        case 1128323: return launch_cfg<AMODE, 128, 32, 3, 1>(ctx, p);
        case 1128643: return launch_cfg<AMODE, 128, 64, 3, 1>(ctx, p);
        case 1064643: return launch_cfg<AMODE, 64, 64, 3, 1>(ctx, p);
        case 1064324: return launch_cfg<AMODE, 64, 32, 4, 1>(ctx, p);
        case 1064323: return launch_cfg<AMODE, 64, 32, 3, 1>(ctx, p);
        // diagnostic (stamped) builds of the plain loop: +3000000
        case 3256642: return launch_cfg<AMODE, 256, 64, 2, 2>(ctx, p);
        case 3128643: return launch_cfg<AMODE, 128, 64, 3, 2>(ctx, p);
        case 3128642: return launch_cfg<AMODE, 128, 64, 2, 2>(ctx, p);
        case 3256324: return launch_cfg<AMODE, 256, 32, 4, 2>(ctx, p);
        default: break;
    }
#endif
    reid_set_error("gemm_f16: no build of tile configuration %d in this library (experiment configurations need -DREID_EXPERIMENTS)", cfg);
    return REID_ERR_ARG;
}

}  // namespace

// strided 3x3 and 1x1 convolutions of the "fp32-class" mode: one build (128-wide tile, BK 32, three stages: two blocks per CU)
int launch_gemm_f16_split(reid_ctx* ctx, const Gemm16Params& p_in, int kind, double flops, double bytes) {
    Gemm16Params p = p_in;
    p.fault = ctx->fault;
    p.general_epi = !ctx->split_lean_epi;
    ARG_CHECK(p.M > 0 && p.M % 128 == 0 && p.N % 128 == 0 && p.K % 32 == 0 && p.ldb % 8 == 0 && p.C32 && p.zero_page &&
              (p.split_terms == 3 || p.split_terms == 4) && p.Cin % (64 * p.split_terms) == 0 && p.K == p.R * p.S * p.Cin);
    prof_begin(ctx, kind, flops, bytes);
    hipLaunchKernelGGL((gemm_f16_kernel<A16_IM2COL, 128, 32, 3, 0, false, true>), dim3(((p.M + 255) / 256) * (p.N / 128)), dim3(512), 0,
                       ctx->stream, p);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_gemm_f16(reid_ctx* ctx, int amode, const Gemm16Params& p_in, int kind, double flops, double bytes) {
    Gemm16Params p = p_in;
    p.fault = ctx->fault;
    ARG_CHECK(p.M > 0 && (p.lin || p.M % 128 == 0) && p.N % 64 == 0 && p.K % 32 == 0 && p.ldb % 8 == 0);
    if (amode == A16_IM2COL) ARG_CHECK(p.Cin % 32 == 0 && p.K == p.R * p.S * p.Cin && p.zero_page);
    if (amode == A16_STEM) ARG_CHECK((p.K == 256 || p.K == 224) && p.Hp >= p.H + 6 && p.Wp >= p.W + 8);
    if (amode == A16_DENSE) ARG_CHECK(p.lda % 8 == 0);
    prof_begin(ctx, kind, flops, bytes);
    int st = REID_ERR_ARG;
    if (amode == A16_IM2COL) st = launch_any<A16_IM2COL>(ctx, p);
    else if (amode == A16_STEM) st = launch_any<A16_STEM>(ctx, p);
    else if (amode == A16_DENSE) st = launch_any<A16_DENSE>(ctx, p);
    else reid_set_error("launch_gemm_f16: unsupported amode %d", amode);
    prof_end(ctx);
    return st;
}
