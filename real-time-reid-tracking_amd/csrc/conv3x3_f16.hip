// 3x3 / stride-1 / pad-1 convolution, fp16 storage, fp32 accumulate, with the input HALO TILE kept in LDS.
//
// Why: the implicit-GEMM kernel (gemm_f16.hip) re-gathers the A operand from global memory for each of the nine taps.
// Measured (tools/bench_feed.py, tools/diag_gemm_f16.py): a CU's L1/TA path delivers ~30 B/clk (67 GB/s from L2, 28 GB/s
// from beyond), so a 256x256x64 tile needs ~2100 cycles of operand feed for 2048 cycles of MFMA, a 256x64 tile (Cout = 64)
// is feed-capped at ~40 % of the MFMA peak, and the DMA issue alone costs ~900 cycles per wave per K-tile.
// Here a block owns 256 output pixels (full-width rows of one image, or two whole 16x8 images), loads the (rows+2) x (W+2)
// halo of its input ONCE per 64-channel chunk and reads the A fragments of all nine taps from it: the A traffic drops ~6x and
// the per-tile feed falls to 13-21 KB (from 40-64 KB), below what the MFMAs of the tile cost.
//
// LDS: two halo buffers (chunk c and c+1) + a 3-slot ring of weight tiles [BN][64] (one tap of one chunk each).
// Both images are lane-linear DMA targets with the source-side XOR swizzle of gemm_f16.hip (128-B rows: chunk c of pixel p at
// position c ^ ((p >> 1) & 7)).  The MFMA row -> pixel map is chosen per geometry so that the 16 lanes of every
// ds_read_b128 group address 16 pixels with distinct (halo index mod 16) at EVERY tap (a tap only shifts the index):
//   W = 32: a 32-row MFMA tile = one image row;  W = 16: one lane group = one image row;
//   W = 8 : one lane group = rows y and y+4 (halo pitch 10: 40 = 8 mod 16).
#include "reid_internal.h"
#include "conv3x3_geom.h"
#include <type_traits>

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(uintptr_t)(p))
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

template <int TW, int IMGS, int BN, int LW, bool SPLIT = false, bool PAIR = false>
__global__ __launch_bounds__(LW ? 768 : 512) void conv3x3_f16_kernel(const Gemm16Params p) {
    static_assert(!PAIR || (SPLIT && LW == 1 && BN == 128), "PAIR: fp32-class build, loader waves, 128-wide tiles");
    constexpr int TH = 256 / (IMGS * TW);            // tile rows per image
    constexpr int WP = TW + 2, HP = TH + 2;          // halo pitch / rows
    constexpr int NPX = IMGS * HP * WP;              // halo pixels per block
    constexpr int NPI = (NPX + 7) / 8;               // halo DMA instructions (8 pixels x 128 B each)
    constexpr int HPW = (NPI + 7) / 8;               // halo instructions per wave (<= 6)
    constexpr int HALO_BYTES = NPI * 1024;
    constexpr int B_BYTES = BN * 128;
    constexpr int BJ = BN / 64;                      // B DMA instructions per wave per tile (BN/8 instructions, 8 waves)
    constexpr int TN = BN / 64;                      // wave tile 64 x BN/2 -> TN 32-col MFMA tiles
    constexpr int TM = 2;
    // Cout = 64 tiles (layer 1) with loader waves: THREE taps per block barrier (a tile of 8 MFMAs per wave per barrier spent more
    // time at the barrier than in the matrix pipe: 507 + 833 cycles per 256 of MFMA, tools/diag_conv_f16.py); the weight ring
    // then holds three groups of three taps
    // (Cout >= 128 tiles have LDS for a fourth weight tile only: TWO taps per barrier over a ring of two pairs ran 4-6 % faster
    // in the 256-crop microbenchmark and 3 % slower in the 1024-crop pass - dropped.)
    constexpr int TPB = (LW && BN == 64 && 2 * HALO_BYTES + 9 * B_BYTES <= 160 * 1024) ? 3 : 1;
    constexpr int NSLOT = PAIR ? 4 : 3 * TPB;
    static_assert(2 * HALO_BYTES + NSLOT * B_BYTES <= 160 * 1024, "LDS budget");
    static_assert(HPW <= 6, "halo pieces are issued one per tap");
    constexpr int NLW = 4;                            // loader waves (LW == 1): waves 8..11 issue every DMA piece
    constexpr int BPL = BN / 8 / NLW;                 // weight pieces per loader wave per tile
    static_assert(6 * 8 >= NPI, "halo pieces of the next chunk are issued over taps 0..7, six per tap");
    __shared__ __attribute__((aligned(16))) char lds[2 * HALO_BYTES + NSLOT * B_BYTES];
    char* halo = lds;
    char* ring = lds + 2 * HALO_BYTES;

    unsigned long long t_entry = 0, t_loop_end = 0;
    if (p.diag) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) :: "memory");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int nnt = p.N / BN;
    const int nwg = gridDim.x;
    // Split-K (few output tiles - a tracking frame): SK blocks share one output tile, each summing Cin / SK of the input
    // channels; the one that finishes last adds the fp32 partials in split order (deterministic) and runs the epilogue.
    const int SK = p.split_k > 1 ? p.split_k : 1;
    int mtile, ntile, tile_id, ksplit;
    {   // XCD-aware, bijective block remap (blocks b and b+8 share an XCD)
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        tile_id = L / SK;
        ksplit = L - tile_id * SK;
        mtile = tile_id / nnt;
        ntile = tile_id - mtile * nnt;
    }
    const int n_blk = ntile * BN;
    // block -> (first image, first row): M rows [256*mtile, +256) in natural (image, y, x) order
    const int tiles_per_img = p.H / TH;              // IMGS == 2 -> 1
    const int img0 = IMGS == 2 ? mtile * 2 : mtile / tiles_per_img;
    const int y0 = IMGS == 2 ? 0 : (mtile - img0 * tiles_per_img) * TH;
    const int n_img = p.M / (p.H * p.W);

    // ---- A fragment addressing: halo pixel index of this lane's row in tile a at tap (0,0)
    int hp0[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        int im, y, x;
        row_to_pixel<TW, IMGS>(wm, a, li, im, y, x);
        hp0[a] = im * HP * WP + y * WP + x;          // tap (r,s) adds r*WP + s  (halo origin = pixel (-1,-1))
    }
    // ---- B descriptors: wave-instruction j fills ring rows (wave*BJ + j)*8 .. +8
    int b_chunk[BJ];
    long long b_base[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = (wave * BJ + j) * 8 + (lane >> 3);
        b_chunk[j] = (lane & 7) ^ ((row >> 1) & 7);
        b_base[j] = (long long)(n_blk + row) * p.ldb;
    }
    // ---- halo descriptors: piece q = wave + 8*tap covers halo pixels q*8 .. +8 (lane/8), chunk position lane%8
    auto issue_halo_piece = [&](int q, int chunk, int buf) {
        const int hp = q * 8 + (lane >> 3);
        const int im = hp / (HP * WP), rem = hp - im * (HP * WP);
        const int hy = rem / WP, hx = rem - hy * WP;
        const int gy = y0 - 1 + hy, gx = hx - 1, gi = img0 + im;
        const bool ok = hp < NPX && gi < n_img && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const int c = (lane & 7) ^ ((hp >> 1) & 7);
        // SPLIT: p.Cin = 3C virtual channels over a tensor of 2C ([xh | xl']): chunks of the last third read xh again
        // (split_terms = 4 adds the xl.wl product: four thirds -> [xh | xl' | xh | xl'] against [wh 2^11 | wh | wl' | wl])
        const int a_cin = SPLIT ? p.Cin / p.split_terms * 2 : p.Cin;
        const int a_chunk = SPLIT ? chunk % (a_cin / 64) : chunk;
        const f16* src = ok ? p.A + (((long long)gi * p.H + gy) * p.W + gx) * a_cin + a_chunk * 64 + c * 8 : p.zero_page;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(halo + buf * HALO_BYTES + q * 1024), 16, 0, 0);
    };
    const int nchunk = p.Cin / 64 / SK;               // chunks this block sums: [chunk0, chunk0 + nchunk)
    const int chunk0 = ksplit * nchunk;
    auto issue_b_piece = [&](int t, int slot, int j) {   // tile t = (chunk - chunk0, tap): weights [Cout][(tap, channel)]
        const int chunk = chunk0 + t / 9, tap = t - (t / 9) * 9;
        const int k0 = tap * p.Cin + chunk * 64;
        __builtin_amdgcn_global_load_lds(GPTR(p.B + b_base[j] + k0 + b_chunk[j] * 8),
                                         LPTR(ring + slot * B_BYTES + (wave * BJ + j) * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int t, int slot) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) issue_b_piece(t, slot, j);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int nt = nchunk * 9;
    const int chunk_end = chunk0 + nchunk;
    const int b_row_off = (wn * (BN / 2) + li) * 128;
    const int b_swz = (li >> 1) & 7;

    const bool is_loader = LW && wave >= 8;
    auto compute_tile = [&](int chunk, int tap, int slot) {
        const char* As = halo + (chunk & 1) * HALO_BYTES;
        const char* Bs = ring + slot * B_BYTES;
        const int r = tap / 3, s = tap - r * 3;
        int a_off[TM], a_swz[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int hp = hp0[a] + r * WP + s;
            a_off[a] = hp * 128;
            a_swz[a] = (hp >> 1) & 7;
        }
        // the fragments of k-step kk + 1 are read before the MFMAs of step kk are issued (two register sets, the order pinned):
        // left to itself the compiler reads three of the four fragments right before the s_waitcnt of the step that needs them
        half8 af[2][TM], bf[2][TN];
        auto read_step = [&](int kk, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < TM; ++a) af[buf][a] = *(const half8*)(As + a_off[a] + (((kk * 2 + lh) ^ a_swz[a]) * 16));
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[buf][b] = *(const half8*)(Bs + b_row_off + b * 32 * 128 + (((kk * 2 + lh) ^ b_swz) * 16));
        };
        read_step(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) read_step(kk + 1, (kk + 1) & 1);
            if (p.frag_ahead) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk & 1][a], bf[kk & 1][b], acc[a][b], 0, 0, 0);
            if (p.frag_ahead) __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (PAIR) {
        // fp32-class arithmetic, operand-sharing order.  x.w = xh.wh + (xl'.wh + xh.wl') 2^-11 as ONE accumulation over the weight
        // parts [wh 2^11 | wh | wl'] (Gemm16Params) - but instead of three passes over the virtual channels (xh, xl', xh again: three
        // halo loads per 64 real channels, one weight tile and 16 MFMAs per wave and barrier) every real chunk c is two PHASES:
        //   phase X, halo = xh_c : nine steps, step = tap t against BOTH the wh 2^11 and the wl' tile (the A fragments are read once
        //                          for the two: 6 fragment reads per 8 MFMAs instead of 8);
        //   phase L, halo = xl'_c: five steps, step = taps (2j, 2j+1) against their wh tiles (the ninth tap alone).
        // 14 barriers per real chunk instead of 27, 32 MFMAs per wave between two barriers instead of 16, two halo loads instead of
        // three.  LDS: xh in halo buffer 0, xl' in buffer 1, the weight ring = two stages of two tiles (the step being computed /
        // the step being loaded).
        // What was measured on the way (round 4, tools/conv_split_ablate.py, layer 4 at 1024 crops, 1.85 ms in the three-pass order):
        // with all-zero operands the three-pass kernel takes 1.48 ms - a fifth of its time is the chip holding its clock down under
        // load; a variant of this loop whose weight ring ran THREE steps ahead (32-channel tiles, counted vmcnt) was 13 % slower:
        // the cost of the DMA stream (0.4-0.6 ms, the same with one step or three steps of lead) is not its latency.
        const int Creal = p.Cin / 3;                       // weights per tap: [wh 2^11 (Creal) | wh | wl']
        const int C64 = Creal / 64;
        const int ncr = C64 / SK, cr0 = ksplit * ncr, cr_end = cr0 + ncr;
        const int nsteps = ncr * 14;
        if (is_loader) {
            if (p.loader_prio) __builtin_amdgcn_s_setprio(3);
            const int lw = wave - 8;
            auto issue_tile = [&](int c, int tap, int part, int slot) {
                const int k0 = tap * p.Cin + part * Creal + c * 64;
#pragma unroll
                for (int j = 0; j < BPL; ++j) {
                    const int inst = lw * BPL + j;
                    const int row = inst * 8 + (lane >> 3);
                    const int cc = (lane & 7) ^ ((row >> 1) & 7);
                    __builtin_amdgcn_global_load_lds(GPTR(p.B + (long long)(n_blk + row) * p.ldb + k0 + cc * 8),
                                                     LPTR(ring + slot * B_BYTES + inst * 1024), 16, 0, 0);
                }
            };
            auto issue_step = [&](int u) {
                const int c = cr0 + u / 14, st = u - (u / 14) * 14, base = (u & 1) * 2;
                if (st < 9) {
                    issue_tile(c, st, 0, base);
                    issue_tile(c, st, 2, base + 1);
                } else {
                    const int t0 = 2 * (st - 9);
                    issue_tile(c, t0, 1, base);
                    if (t0 + 1 < 9) issue_tile(c, t0 + 1, 1, base + 1);
                }
            };
            constexpr int PX = (NPI + 8) / 9, PL = (NPI + 4) / 5;      // halo pieces of the next phase per step of this one
            for (int q = lw; q < NPI; q += NLW) issue_halo_piece(q, cr0, 0);
            issue_step(0);
            const int abl = p.ablate;   // experiments (reid_debug_conv_split): 1 no weight DMA, 2 no halo DMA after the prologue
            for (int u = 0; u < nsteps; ++u) {
                WAIT_VMCNT(0);        // everything this loader issued one step ago (the step's weights, halo pieces) has landed
                RAW_BARRIER();
                if (u + 1 < nsteps && !(abl & 1)) issue_step(u + 1);
                const int c = cr0 + u / 14, st = u - (u / 14) * 14;
                if (abl & 2) continue;
                if (st < 9) {         // xl'_c into buffer 1 (last read in phase L of chunk c - 1: every wave is past it)
                    const int q1 = (st + 1) * PX < NPI ? (st + 1) * PX : NPI;
                    for (int q = st * PX + lw; q < q1; q += NLW) issue_halo_piece(q, C64 + c, 1);
                } else if (c + 1 < cr_end) {   // xh_{c+1} into buffer 0 (last read in step 8 of this chunk)
                    const int j = st - 9, q1 = (j + 1) * PL < NPI ? (j + 1) * PL : NPI;
                    for (int q = j * PL + lw; q < q1; q += NLW) issue_halo_piece(q, c + 1, 0);
                }
            }
            return;
        }
        // compute waves.  The fragment reads are inline asm and their waits are counted by hand: left to hipcc the waits in this loop
        // come out as s_waitcnt lgkmcnt(0) (the wait in front of the MFMAs of k-step kk then also waits for the fragments just
        // requested for kk + 1: reads and MFMAs serialise - that build ran at HALF the speed).  A group = the fragments of one k-step
        // (2 A + 4 B in phase X, 2 A + 2 B in phase L), two register sets; group g + 1 is requested before the MFMAs of group g
        // issue.  The block barrier of step u + 1 sits in front of the LAST MFMA group of step u, as soon as every fragment of step u
        // has been read: the first fragments of step u + 1 are then in flight under those MFMAs.
        const unsigned halo32 = (unsigned)(uintptr_t)halo, ring32 = (unsigned)(uintptr_t)ring;
        unsigned bx[4];                                   // B fragment address of k-step kk inside a weight tile (column block 0)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) bx[kk] = ring32 + b_row_off + (((kk * 2 + lh) ^ b_swz) * 16);
#define LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
        // A fragment addresses of tap t in halo buffer `buf`: per MFMA row tile a and k-step kk
        auto a_addrs = [&](int buf, int tap, unsigned (&aa)[TM][4]) __attribute__((always_inline)) {
            const int r = tap / 3, sx = tap - r * 3;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int hp = hp0[a] + r * WP + sx;
                const unsigned base = halo32 + buf * HALO_BYTES + hp * 128;
                const int swz = (hp >> 1) & 7;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) aa[a][kk] = base + (((kk * 2 + lh) ^ swz) * 16);
            }
        };
        static_assert(TN == 2, "fragment offsets below assume two 32-column blocks per wave");
        for (int c = 0; c < ncr; ++c) {
            {   // ---- phase X: nine steps, tap st of xh_c against the weight tiles (wh 2^11, wl') of ring stage st & 1
                half8 fa[2][TM], fb[2][2 * TN];
                unsigned aa[TM][4];
                auto rd = [&](int set, int kk, unsigned soff) __attribute__((always_inline)) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) LDS_READ(fa[set][a], aa[a][kk], 0);
                    const unsigned ba = bx[kk] + soff;
                    LDS_READ(fb[set][0], ba, 0);
                    LDS_READ(fb[set][1], ba, 4096);
                    LDS_READ(fb[set][2], ba, B_BYTES);
                    LDS_READ(fb[set][3], ba, B_BYTES + 4096);
                };
                auto mm = [&](int set) __attribute__((always_inline)) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int a = 0; a < TM; ++a)
#pragma unroll
                            for (int b = 0; b < TN; ++b)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][a], fb[set][t * TN + b], acc[a][b], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                RAW_BARRIER();
                a_addrs(0, 0, aa);
                rd(0, 0, 0u);
#pragma unroll 1
                for (int st = 0; st < 9; ++st) {
                    const unsigned soff = (st & 1) * 2 * B_BYTES;
                    rd(1, 1, soff); LGKM_WAIT(6); mm(0);
                    rd(0, 2, soff); LGKM_WAIT(6); mm(1);
                    rd(1, 3, soff); LGKM_WAIT(6); mm(0);
                    LGKM_WAIT(0);
                    if (p.pair_early && st < 8) {
                        RAW_BARRIER();
                        a_addrs(0, st + 1, aa);
                        rd(0, 0, ((st + 1) & 1) * 2 * B_BYTES);
                    }
                    mm(1);
                    if (!p.pair_early && st < 8) {
                        RAW_BARRIER();
                        a_addrs(0, st + 1, aa);
                        rd(0, 0, ((st + 1) & 1) * 2 * B_BYTES);
                    }
                }
            }
            {   // ---- phase L: nine passes (taps of xl'_c, one wh tile each), two per step (the ninth alone): steps 9 .. 13
                half8 fa[2][TM], fb[2][TN];
                unsigned aa[TM][4];
                auto rd = [&](int set, int kk, unsigned soff) __attribute__((always_inline)) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) LDS_READ(fa[set][a], aa[a][kk], 0);
                    const unsigned ba = bx[kk] + soff;
                    LDS_READ(fb[set][0], ba, 0);
                    LDS_READ(fb[set][1], ba, 4096);
                };
                auto mm = [&](int set) __attribute__((always_inline)) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][a], fb[set][b], acc[a][b], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto slot_off = [&](int q) { return (unsigned)(((((q >> 1) + 1) & 1) * 2 + (q & 1)) * B_BYTES); };   // pass q: step 9 + q / 2
                RAW_BARRIER();
                a_addrs(1, 0, aa);
                rd(0, 0, slot_off(0));
#pragma unroll 1
                for (int q = 0; q < 9; ++q) {
                    const unsigned soff = slot_off(q);
                    rd(1, 1, soff); LGKM_WAIT(4); mm(0);
                    rd(0, 2, soff); LGKM_WAIT(4); mm(1);
                    rd(1, 3, soff); LGKM_WAIT(4); mm(0);
                    LGKM_WAIT(0);
                    const bool late = !p.pair_early && (q & 1);   // a step's barrier after its last MFMA group instead of before it
                    if (q < 8 && !late) {
                        if (q & 1) RAW_BARRIER();          // passes 2 j and 2 j + 1 form one step
                        a_addrs(1, q + 1, aa);
                        rd(0, 0, slot_off(q + 1));
                    }
                    mm(1);
                    if (q < 8 && late) {
                        RAW_BARRIER();
                        a_addrs(1, q + 1, aa);
                        rd(0, 0, slot_off(q + 1));
                    }
                }
            }
        }
#undef LDS_READ
#undef LGKM_WAIT
    } else if constexpr (LW == 0) {
        // prologue: halo of chunk 0, weight tiles 0 and 1
#pragma unroll
        for (int k = 0; k < HPW; ++k)
            if (wave + 8 * k < NPI) issue_halo_piece(wave + 8 * k, chunk0, chunk0 & 1);
        issue_b(0, 0);
        if (nt > 1) issue_b(1, 1);
        int prev_b = nt > 1 ? 1 : 0, prev_h = 0;   // what the previous iteration issued (for the counted wait)
        int slot_c = 0, slot_i = 2, chunk = chunk0, tap = 0;
        for (int t = 0; t < nt; ++t) {
            // everything except what was issued in the previous iteration has landed after this wait
            if (prev_b && prev_h) WAIT_VMCNT(BJ + 1);
            else if (prev_b) WAIT_VMCNT(BJ);
            else if (prev_h) WAIT_VMCNT(1);
            else WAIT_VMCNT(0);
            RAW_BARRIER();
            prev_b = 0;
            prev_h = 0;
            if (t + 2 < nt) { issue_b(t + 2, slot_i); prev_b = 1; }
            if (tap < HPW && chunk + 1 < chunk_end && wave + 8 * tap < NPI) { issue_halo_piece(wave + 8 * tap, chunk + 1, (chunk + 1) & 1); prev_h = 1; }
            compute_tile(chunk, tap, slot_c);
            slot_c = slot_c == 2 ? 0 : slot_c + 1;
            slot_i = slot_i == 2 ? 0 : slot_i + 1;
            if (++tap == 9) { tap = 0; ++chunk; }
        }
    } else {
        // Warp-specialised: waves 8..11 only move data (their DMA issue stalls - ~110 cycles per piece on the L1/TA path -
        // no longer sit in the instruction stream of the MFMA waves), waves 0..7 only read LDS and issue MFMAs.  One block
        // barrier per tile: a loader arrives once its pieces of the NEXT tile have landed, a compute wave once it has
        // finished the current tile, so after barrier t tile t is complete in LDS and the slot of tile t-1 is free.
        if constexpr (TPB == 3) {
            const int nu = nt / 3;                       // iterations: (chunk, group of three taps); nt = 9 * nchunk
            if (is_loader) {
                if (p.loader_prio) __builtin_amdgcn_s_setprio(3);
                const int lw = wave - 8;
                auto loader_group = [&](int u) {         // weights of taps 3g .. 3g+2 of chunk u / 3 into ring slots 3 (u % 3) + j
                    const int ck = chunk0 + u / 3, g = u - (u / 3) * 3;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int k0 = (g * 3 + j) * p.Cin + ck * 64;
                        const int slot = (u % 3) * 3 + j;
#pragma unroll
                        for (int jj = 0; jj < BPL; ++jj) {
                            const int inst = lw * BPL + jj;
                            const int row = inst * 8 + (lane >> 3);
                            const int c = (lane & 7) ^ ((row >> 1) & 7);
                            __builtin_amdgcn_global_load_lds(GPTR(p.B + (long long)(n_blk + row) * p.ldb + k0 + c * 8),
                                                             LPTR(ring + slot * B_BYTES + inst * 1024), 16, 0, 0);
                        }
                    }
                };
                auto wait_all_but = [&](int n) {         // s_waitcnt takes an immediate
                    switch (n) {
                        case 0: WAIT_VMCNT(0); break;   case 1: WAIT_VMCNT(1); break;   case 2: WAIT_VMCNT(2); break;
                        case 3: WAIT_VMCNT(3); break;   case 4: WAIT_VMCNT(4); break;   case 5: WAIT_VMCNT(5); break;
                        case 6: WAIT_VMCNT(6); break;   case 7: WAIT_VMCNT(7); break;   case 8: WAIT_VMCNT(8); break;
                        case 9: WAIT_VMCNT(9); break;   case 10: WAIT_VMCNT(10); break; case 11: WAIT_VMCNT(11); break;
                        case 12: WAIT_VMCNT(12); break; case 13: WAIT_VMCNT(13); break; case 14: WAIT_VMCNT(14); break;
                        default: WAIT_VMCNT(0); break;
                    }
                };
                constexpr int HALF = (NPI + 1) / 2;      // halo pieces of the next chunk: half in each of a chunk's first two iterations
                for (int q = lw; q < NPI; q += NLW) issue_halo_piece(q, chunk0, chunk0 & 1);
                loader_group(0);
                if (nu > 1) loader_group(1);
                int last = nu > 1 ? 3 * BPL : 0;         // pieces in the most recently issued group
                int chunk = chunk0, g = 0;
                for (int u = 0; u < nu; ++u) {
                    // all but the most recent group (the next iteration's weights, halo pieces issued beside them) have landed
                    wait_all_but(last);
                    RAW_BARRIER();
                    last = 0;
                    if (u + 2 < nu) { loader_group(u + 2); last += 3 * BPL; }
                    // SPLIT with 64 real channels (layer 1): virtual chunk 2 is xh again - what chunk 0 left in the same buffer
                    const bool resident = SPLIT && p.Cin == 192 && p.split_terms == 3 && chunk0 == 0 && chunk + 1 == 2;
                    if (g < 2 && chunk + 1 < chunk_end && !resident) {
                        const int q1 = (g + 1) * HALF < NPI ? (g + 1) * HALF : NPI;
                        for (int q = g * HALF + lw; q < q1; q += NLW) { issue_halo_piece(q, chunk + 1, (chunk + 1) & 1); ++last; }
                    }
                    if (++g == 3) { g = 0; ++chunk; }
                }
                return;
            } else {
                int chunk = chunk0, g = 0;
                for (int u = 0; u < nu; ++u) {
                    RAW_BARRIER();
#pragma unroll
                    for (int j = 0; j < 3; ++j) compute_tile(chunk, g * 3 + j, g * 3 + j);     // u % 3 == g: nine taps per chunk
                    if (++g == 3) { g = 0; ++chunk; }
                }
            }
        } else {
        int slot_c = 0, slot_i = 2, chunk = chunk0, tap = 0;
        if (is_loader) {
            if (p.loader_prio) __builtin_amdgcn_s_setprio(3);
            const int lw = wave - 8;
            auto loader_b = [&](int t, int slot) {   // this loader's BPL pieces of weight tile t
                const int ck = chunk0 + t / 9, tp = t - (t / 9) * 9;
                const int k0 = tp * p.Cin + ck * 64;
#pragma unroll
                for (int j = 0; j < BPL; ++j) {
                    const int inst = lw * BPL + j;
                    const int row = inst * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ ((row >> 1) & 7);
                    __builtin_amdgcn_global_load_lds(GPTR(p.B + (long long)(n_blk + row) * p.ldb + k0 + c * 8),
                                                     LPTR(ring + slot * B_BYTES + inst * 1024), 16, 0, 0);
                }
            };
            for (int q = lw; q < NPI; q += NLW) issue_halo_piece(q, chunk0, chunk0 & 1);
            loader_b(0, 0);
            if (nt > 1) loader_b(1, 1);
            int last = nt > 1 ? BPL : 0;   // pieces in the most recently issued group
            for (int t = 0; t < nt; ++t) {
                // all but the most recent group (tile t+1's weights) have landed: tile t and its halo are complete
                if (last == BPL + 2) WAIT_VMCNT(BPL + 2);
                else if (last == BPL + 1) WAIT_VMCNT(BPL + 1);
                else if (last == BPL) WAIT_VMCNT(BPL);
                else WAIT_VMCNT(0);
                RAW_BARRIER();
                last = 0;
                if (t + 2 < nt) { loader_b(t + 2, slot_i); last += BPL; }
                if (tap < 8 && chunk + 1 < chunk_end) {   // halo of the next chunk: six pieces per tap over taps 0..7
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int q = tap * 6 + lw + 4 * k;
                        if (lw + 4 * k < 6 && q < NPI) { issue_halo_piece(q, chunk + 1, (chunk + 1) & 1); ++last; }
                    }
                }
                slot_i = slot_i == 2 ? 0 : slot_i + 1;
                if (++tap == 9) { tap = 0; ++chunk; }
            }
            // Every piece has landed (the wait before the last barrier was vmcnt(0)): the loaders are done.  A finished
            // wave no longer counts towards s_barrier, so the eight compute waves run the epilogue on their own.
            return;
        } else {
            unsigned long long t_bar = 0, t_comp = 0, ta = 0, tb = 0;
            const bool dg = p.diag != nullptr;   // diagnostic stamps (experiments only)
            for (int t = 0; t < nt; ++t) {
                if (dg) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ta) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
                RAW_BARRIER();
                if (dg) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb) :: "memory"); __builtin_amdgcn_sched_barrier(0); t_bar += tb - ta; }
                compute_tile(chunk, tap, slot_c);
                if (dg) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ta) :: "memory"); __builtin_amdgcn_sched_barrier(0); t_comp += ta - tb; }
                slot_c = slot_c == 2 ? 0 : slot_c + 1;
                if (++tap == 9) { tap = 0; ++chunk; }
            }
            if (dg && lane == 0 && blockIdx.x < 64) {
                unsigned long long* d = p.diag + ((long long)blockIdx.x * 8 + wave) * 4;
                d[0] = t_bar; d[1] = t_comp; d[2] = nt; d[3] = ta - t_entry;   // d[3]: kernel entry -> end of the K loop
                t_loop_end = ta;
            }
        }
        }
    }
    __syncthreads();

    if (SK > 1) {
        // partial tile in register order [TM*TN*16][512 lanes]: coalesced both ways
        constexpr int PART = 256 * BN;
        float* part = p.splitk_ws + (long long)tile_id * SK * PART;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    __hip_atomic_store(part + (long long)ksplit * PART + ((a * TN + b) * 16 + e) * 512 + tid, acc[a][b][e], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        // Device-scope (sc1) stores are written through to the point all XCDs share and the loads below bypass this XCD's L2: no
        // L2 write-back / invalidate fence is needed (a __threadfence() pair here flushed the whole L2 240 times per launch and
        // tripled the kernel's duration).  The stores have been acknowledged once vmcnt reaches 0.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = (int*)lds;
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(p.splitk_cnt + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = old == SK - 1;
            if (old == SK - 1) __hip_atomic_store(p.splitk_cnt + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // everyone has arrived: ready for the next launch
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
                        acc[a][b][e] += __hip_atomic_load(part + (long long)sidx * PART + ((a * TN + b) * 16 + e) * 512 + tid, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();   // the flag word is about to be overwritten by the residual tile
    }

    if constexpr (SPLIT) {
        if (p.ablate & 32) return;    // experiment: no epilogue
        // ------------------------------------------------------------------ fp32 epilogue of the SPLIT build: BN scale (x 2^-11)
        // and shift, fp32 residual, ReLU from column relu_from on, fp32 stores, per-128-row column sums - straight from the
        // accumulators (a row of the MFMA tile is 32 consecutive columns: 128-byte segments).  Rows past M only occur in the
        // second 128-row half (M % 128 == 0): their row index is clamped for the residual load and their store is skipped.
        const int ldc = (int)p.ldc;
        const int m_blk = mtile * 256;
        const int m_valid = p.M - m_blk;
        float s1[TN], s2[TN];
        float vmax = 0.f;   // largest packed magnitude (range guard: the accumulators cannot hold a NaN that did not start as an inf)
        // Lean form (every launch of the forward): no per-element predicate, 64-bit address or branch.  M % 128 == 0, so a ragged
        // last tile has exactly 128 rows and they belong to the waves wm 0, 1 (rows 0 .. 127 in every geometry): the other waves skip
        // their loads and stores as a whole.  A row's byte offset splits into a lane part (which of the tile's two row groups the
        // lane half holds: it depends on the lane half and, for the 16- and 8-wide geometries, on the register's quarter) and a
        // uniform part per (a, e): buffer instructions take the first in the vector offset, the second in the scalar offset, the
        // column block's +128 B as the immediate.  The arithmetic per element is the general loop's, operation for operation.
        const bool lean = !p.general_epi && (long long)256 * ldc * 4 < 0x7fffff00ll && (long long)256 * 2 * p.N * 2 < 0x7fffff00ll && (m_valid >= 256 || m_valid == 128);
        if (lean) {
            const bool wave_live = m_valid >= 256 || wm < 2;
            // lane part of the natural row, for registers whose quarter q = e >> 2 is 0 / 3 (x = 0) and 1 / 2 (x = 1)
            const int lrow[2] = {c_row_lane<TW, IMGS>(lh, 0), c_row_lane<TW, IMGS>(lh, 1)};
            const int col0 = n_blk + wn * (BN / 2) + li;
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
                        float r[16];
                        if constexpr (RES) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int x = ((e >> 2) == 1 || (e >> 2) == 2) ? 1 : 0;
                                const int urow = c_row_uniform<TW, IMGS>(wm, a, e);
                                r[e] = wave_live ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, voff[x] + b * 128, urow * ldc * 4, 0)) : 0.f;
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int x = ((e >> 2) == 1 || (e >> 2) == 2) ? 1 : 0;
                            const int urow = c_row_uniform<TW, IMGS>(wm, a, e);
                            float v = acc[a][b][e] * cs + sh;
                            v += RES ? r[e] : 0.f;           // as the general loop: + 0 when there is no residual
                            v = fmaxf(v, lo);
                            if (wave_live) {
                                t1 += v;
                                t2 += v * v;
                                if (pk) {
                                    vmax = fmaxf(vmax, fabsf(v));
                                    const f16 hv = (f16)v;
                                    const f16 lv = (f16)((v - (float)hv) * 2048.0f);
                                    // (pairing neighbouring lanes' values into one dword store per lane - DPP quad_perm - was measured: slower)
                                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), k_rs, koff[x] + b * 64, urow * 2 * p.N * 2, 0);
                                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, lv), k_rs, koff[x] + b * 64 + p.N * 2, urow * 2 * p.N * 2, 0);
                                } else {
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
                const int lcol = wn * (BN / 2) + b * 32 + li;
                const int col = n_blk + lcol;
                const float cs = (p.col_scale ? p.col_scale[col] : 1.f) * p.acc_scale;
                const float sh = p.col_scale ? p.col_shift[col] : 0.f;
                const float lo = (p.relu && col >= p.relu_from) ? 0.f : -INFINITY;
                const bool pk = p.pack16 && col >= p.pack_from;
                float t1 = 0.f, t2 = 0.f;
    #pragma unroll
                for (int a = 0; a < TM; ++a) {
                    float r[16];
                    int rowv[16];
    #pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        rowv[e] = c_row_natural<TW, IMGS>(wm, a, e, lh);
                        const int rc = rowv[e] < m_valid ? rowv[e] : 0;
                        r[e] = p.res32 ? p.res32[(long long)(m_blk + rc) * ldc + col] : 0.f;
                    }
    #pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float v = acc[a][b][e] * cs + sh;
                        v += r[e];
                        v = fmaxf(v, lo);
                        if (rowv[e] < m_valid) {
                            t1 += v;
                            t2 += v * v;
                            if (pk) {   // [yh | yl'] for the next convolution's loader (uniform per 32-column tile: pack_from % 32 == 0)
                                vmax = fmaxf(vmax, fabsf(v));
                                const f16 hv = (f16)v;
                                f16* dst = p.pack16 + (long long)(m_blk + rowv[e]) * 2 * p.N + col;
                                dst[0] = hv;
                                dst[p.N] = (f16)((v - (float)hv) * 2048.0f);
                            } else if (!(p.ablate & 64)) {
                                p.C32[(long long)(m_blk + rowv[e]) * ldc + col] = v;
                            } else {
                                asm volatile("" ::"v"(v));
                            }
                        }
                    }
                }
                s1[b] = t1;
                s2[b] = t2;
            }
        }
        if (p.fault && !(vmax < 65504.f)) p.fault[0] = 1;   // a packed activation f16 cannot hold: the context reports it
        if (p.stats) {   // per 128 natural rows: waves wm 0,1 own rows 0..127, wm 2,3 rows 128..255 in every geometry
            float* stat_lds = (float*)lds;  // [4][BN][2]
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int lcol = wn * (BN / 2) + b * 32 + li;
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
                float* o = p.stats + ((long long)(mtile * 2 + half) * p.N + n_blk + c) * 2;
                o[0] = stat_lds[((half * 2) * BN + c) * 2 + 0] + stat_lds[((half * 2 + 1) * BN + c) * 2 + 0];
                o[1] = stat_lds[((half * 2) * BN + c) * 2 + 1] + stat_lds[((half * 2 + 1) * BN + c) * 2 + 1];
            }
        }
        return;
    }
    // ------------------------------------------------------------------ epilogue (fp32 math, f16 stores)
    const int ldc = (int)p.ldc;
    const int m_blk = mtile * 256;
    f16* Cb = p.C + (long long)m_blk * ldc + n_blk;
    const f16* Rb = p.residual ? p.residual + (long long)m_blk * ldc + n_blk : nullptr;
    const int m_valid = p.M - m_blk;
    if (Rb) {   // residual tile [256 natural-order rows][BN] via DMA into the free LDS
        constexpr int RROW = BN * 2, RCH = RROW / 16, RRPI = 1024 / RROW, RJ = 256 / RRPI / 8;
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int inst = wave * RJ + j;
            const int row = inst * RRPI + lane / RCH;
            const f16* src = row < m_valid ? Rb + (long long)row * ldc + (lane % RCH) * 8 : p.zero_page;
            __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(lds + inst * 1024), 16, 0, 0);
        }
        __syncthreads();
    }
    const f16* Rl = (const f16*)lds;
    constexpr int NTHR = 512;   // the loader waves of the LW variant have exited
    // pass 1: BN / residual / ReLU in fp32, in place in the accumulators; per-column partial sums.  Straight-line and
    // branch-free: rows past m_valid (ragged last tile) all belong to the waves of the second 128-row half, whose
    // statistics and stores are skipped as a whole, so no per-element predicate is needed; flags become operands.
    float s1[TN], s2[TN];
    const float lo = p.relu ? 0.f : -INFINITY;
    auto pass1 = [&](auto with_res) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int lcol = wn * (BN / 2) + b * 32 + li;
            const int col = n_blk + lcol;
            const float cs = p.col_scale ? p.col_scale[col] : 1.f;
            const float sh = p.col_scale ? p.col_shift[col] : 0.f;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
#pragma unroll
                for (int eg = 0; eg < 4; ++eg) {   // four rows at a time keeps the live set inside the 168-VGPR budget
                    float r[4];
                    if constexpr (decltype(with_res)::value) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] = (float)Rl[c_row_natural<TW, IMGS>(wm, a, eg * 4 + e, lh) * BN + lcol];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[a][b][eg * 4 + e] * cs + sh;
                        if constexpr (decltype(with_res)::value) v += r[e];
                        v = fmaxf(v, lo);
                        t1 += v;
                        t2 += v * v;
                        acc[a][b][eg * 4 + e] = v;
                    }
                }
            }
            s1[b] = t1;
            s2[b] = t2;
        }
    };
    if (Rb) pass1(std::true_type{});
    else pass1(std::false_type{});
    unsigned long long te1 = 0, te2 = 0, te3 = 0, te4 = 0;
#define ESTAMP(v) if (LW && p.diag) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
    ESTAMP(te1);
    __syncthreads();   // every wave is done with the residual tile
    if (p.stats) {   // per 128 natural rows: waves wm 0,1 own rows 0..127, wm 2,3 rows 128..255 in every geometry
        float* stat_lds = (float*)lds;  // [4][BN][2]
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int lcol = wn * (BN / 2) + b * 32 + li;
            const float t1 = s1[b] + __shfl_xor(s1[b], 32);
            const float t2 = s2[b] + __shfl_xor(s2[b], 32);
            if (lh == 0) {
                stat_lds[(wm * BN + lcol) * 2 + 0] = t1;
                stat_lds[(wm * BN + lcol) * 2 + 1] = t2;
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * BN; t += NTHR) {
            const int half = t / BN, c = t - half * BN;
            if (half * 128 >= m_valid) continue;
            float* o = p.stats + ((long long)(mtile * 2 + half) * p.N + n_blk + c) * 2;
            o[0] = stat_lds[((half * 2) * BN + c) * 2 + 0] + stat_lds[((half * 2 + 1) * BN + c) * 2 + 0];
            o[1] = stat_lds[((half * 2) * BN + c) * 2 + 1] + stat_lds[((half * 2 + 1) * BN + c) * 2 + 1];
        }
        __syncthreads();
    }
    ESTAMP(te2);
    // pass 2: output tile in natural row order through LDS, then whole 16-byte pieces of a row per lane (see gemm_f16.hip)
    f16* tile = (f16*)lds;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int lcol = wn * (BN / 2) + b * 32 + li;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) tile[c_row_natural<TW, IMGS>(wm, a, e, lh) * BN + lcol] = (f16)acc[a][b][e];
    }
    ESTAMP(te3);
    __syncthreads();
    constexpr int C8 = BN / 8;
    for (int idx = tid; idx < 256 * C8; idx += NTHR) {
        const int row = idx / C8, c8 = idx - row * C8;
        if (row < m_valid) *(half8*)(Cb + (long long)row * ldc + c8 * 8) = *(const half8*)(tile + row * BN + c8 * 8);
    }
    if (LW && p.diag && lane == 0 && blockIdx.x < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ESTAMP(te4);
        if (wave == 0) {
            unsigned long long* d2 = p.diag + 64 * 8 * 4 + blockIdx.x * 8;
            d2[0] = te1 - t_loop_end; d2[1] = te2 - te1; d2[2] = te3 - te2; d2[3] = te4 - te3;   // pass1 | sync+stats | LDS writes | sync+stores+drain
        }
    }
}

template <int TW, int IMGS, int LW, bool SPLIT = false>
int launch_geom(reid_ctx* ctx, const Gemm16Params& p0) {
    Gemm16Params p = p0;
    p.fault = ctx->fault;
    p.general_epi = !ctx->split_lean_epi;
    p.pair_early = ctx->split_pair == 2;
    p.loader_prio = ctx->f16_loader_prio == 2 || (ctx->f16_loader_prio == 1 && SPLIT);
    p.frag_ahead = ctx->f16_frag_ahead;
    const int nmt = (p.M + 255) / 256;
    const int threads = LW ? 768 : 512;
    // few M tiles (a tracking frame): 64-wide N tiles put twice as many blocks on the chip ...
    const int nchunk = p.Cin / 64;
    auto splitk = [&](int tiles, int bn, int sk) -> int {
        if (sk > 1) {
            float* ws;
            int* cnt;
            bool fresh = ctx->ws.find("conv16.splitk_cnt") == ctx->ws.end();
            REID_TRY(ctx_ws(ctx, "conv16.splitk_ws", (size_t)tiles * sk * 256 * bn * sizeof(float), (void**)&ws));
            REID_TRY(ctx_ws(ctx, "conv16.splitk_cnt", 256 * sizeof(int), (void**)&cnt));
            if (fresh) HIP_TRY(hipMemsetAsync(cnt, 0, 256 * sizeof(int), ctx->stream));
            p.split_k = sk; p.splitk_ws = ws; p.splitk_cnt = cnt;
        }
        return REID_OK;
    };
    const int tiles128 = p.N % 128 == 0 ? nmt * (p.N / 128) : 0;
#ifdef REID_EXPERIMENTS
    constexpr bool CAN_PAIR = SPLIT && LW == 1;
#else
    constexpr bool CAN_PAIR = false;   // the operand-sharing PAIR order (switch split_pair: +1.1 % / -0.9 %, another summation order): experiment builds only
#endif
    const bool pair = CAN_PAIR && ctx->split_pair && p.split_terms == 3;   // the operand-sharing order of the fp32-class products
    if (tiles128 > 128) {
        if constexpr (CAN_PAIR) {
            if (pair) hipLaunchKernelGGL((conv3x3_f16_kernel<TW, IMGS, 128, LW, SPLIT, true>), dim3(tiles128), dim3(threads), 0, ctx->stream, p);
        }
        if (!pair) hipLaunchKernelGGL((conv3x3_f16_kernel<TW, IMGS, 128, LW, SPLIT>), dim3(tiles128), dim3(threads), 0, ctx->stream, p);
    } else {
        // Few output tiles (a tracking frame): the launch leaves CUs idle and its K loop is latency-bound at ~0.65 us per (chunk, tap)
        // tile of a 64-wide block and ~1.3 us of a 128-wide one, so its length is what the launch costs.  Split the input channels
        // over sk blocks per output tile and pick, among 64- and 128-wide tiles and sk = 1 .. 4, the form with the shortest
        // rounds x K-loop (a block owns a CU: 256 at a time).  128-wide tiles (half the barriers, 1.0 instead of 1.5 fragment reads
        // per MFMA) need a long K loop to split (layer 4).  sk = 3 is what the Poisson(30) frames of 17-23 and 33-42 crops were
        // missing: 36 crops took 1.24 ms against 0.83 ms for 30 (layer 4 as 144 unsplit blocks) - now 1.0x.
        const int tiles64 = nmt * (p.N / 64);
        int best_bn = 64, best_sk = 1;
        double best = 1e30;
        auto consider = [&](int bn, int tiles, int sk, double bias) {
            if (tiles <= 0 || nchunk % sk != 0 || (sk > 1 && (!ctx->f16_split_k || tiles * sk > 256))) return;
            const double cost = (double)((tiles * sk + 255) / 256) * (nchunk / sk) * 9 * (bn == 128 ? 1.3 : 0.65) + (sk > 1 ? 2.0 : 0.0) + bias;
            if (cost < best) { best = cost; best_bn = bn; best_sk = sk; }
        };
        // (ties go to the forms of earlier rounds, so that a frame size they served keeps its summation order: 128-wide x 4, then
        // 64-wide with sk a power of two)
        if (ctx->f16_wide_splitk && nchunk >= 16 && tiles128 >= 48) consider(128, tiles128, 4, -0.003);
        for (int sk = 4; sk >= 1; sk >>= 1) consider(64, tiles64, sk, -0.002 + 0.0001 * sk);
        consider(64, tiles64, 3, 0.0);
        // exactly 128 wide tiles (a pass of 64 crops - the library's default pass size - in layers 2 and 4): as 256 tiles of 64 columns
        // every CU has a block instead of every other one; unsplit, so each output still sums its K range in the same order and the
        // results are the 128-wide launch's bit for bit
        if (tiles128 == 128) consider(128, tiles128, 1, 0.0005);
        if (ctx->f16_wide_splitk && nchunk >= 12 && tiles128 >= 48) {   // (layer 4's first convolution: 12 chunks)
            consider(128, tiles128, 4, 0.001);
            consider(128, tiles128, 3, 0.001);
            consider(128, tiles128, 2, 0.001);
        }
        const int tiles = best_bn == 128 ? tiles128 : tiles64;
        REID_TRY(splitk(tiles, best_bn, best_sk));
        if (best_bn == 128) {
            if constexpr (CAN_PAIR) {
                if (pair) hipLaunchKernelGGL((conv3x3_f16_kernel<TW, IMGS, 128, LW, SPLIT, true>), dim3(tiles * best_sk), dim3(threads), 0, ctx->stream, p);
            }
            if (!pair) hipLaunchKernelGGL((conv3x3_f16_kernel<TW, IMGS, 128, LW, SPLIT>), dim3(tiles * best_sk), dim3(threads), 0, ctx->stream, p);
        } else {
            hipLaunchKernelGGL((conv3x3_f16_kernel<TW, IMGS, 64, LW, SPLIT>), dim3(tiles * best_sk), dim3(threads), 0, ctx->stream, p);
        }
    }
    LAUNCH_CHECK();
    return REID_OK;
}

}  // namespace

// true when the halo kernel covers this convolution (3x3, stride 1, pad 1, Cin % 64 == 0, one of the three map sizes)
bool conv3x3_f16_supported(const Gemm16Params& p) {
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || p.Cin % 64 != 0 || p.N % 64 != 0) return false;
    return (p.W == 32 && p.H % 8 == 0) || (p.W == 16 && p.H % 16 == 0) || (p.W == 8 && p.H == 16);
}

int launch_conv3x3_f16(reid_ctx* ctx, const Gemm16Params& p, int kind, double flops, double bytes) {
    ARG_CHECK(conv3x3_f16_supported(p) && p.zero_page && p.ldb % 8 == 0 && p.M % 128 == 0);
    prof_begin(ctx, kind, flops, bytes);
    int st;
    if (ctx->f16_loader_waves) {
        if (p.W == 32) st = launch_geom<32, 1, 1>(ctx, p);
        else if (p.W == 16) st = launch_geom<16, 1, 1>(ctx, p);
        else st = launch_geom<8, 2, 1>(ctx, p);
    } else {
#ifdef REID_EXPERIMENTS       // the builds without loader waves (switch f16_loader_waves = 0: 0.806 against 0.856 of peak): experiment builds only
        if (p.W == 32) st = launch_geom<32, 1, 0>(ctx, p);
        else if (p.W == 16) st = launch_geom<16, 1, 0>(ctx, p);
        else st = launch_geom<8, 2, 0>(ctx, p);
#else
        reid_set_error("conv3x3_f16: the builds without loader waves need a library made with -DREID_EXPERIMENTS");
        st = REID_ERR_ARG;
#endif
    }
    prof_end(ctx);
    return st;
}

// ---- "fp32-class" convolutions on the f16 matrix pipe (Gemm16Params: SPLIT build) ---------------------------------------------
namespace {
// fp32 [rows][C] -> f16 [rows][2C]: [xh | xl'], xh = f16(x), xl' = f16((x - xh) * 2^11); eight channels per thread
// scale (may be null): a device scalar every value is multiplied by first (a power of two: knn_wide.hip normalises its operands)
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ x, long long rows, int C, f16* __restrict__ out,
                                                         int* __restrict__ fault, const float* __restrict__ scale) {
    const int c8 = C >> 3;
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= rows * c8) return;
    const long long row = i / c8;
    const int c = (int)(i - row * c8) * 8;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 a = *(const f32x4*)(x + row * C + c), b = *(const f32x4*)(x + row * C + c + 4);
    const float sc = scale ? scale[0] : 1.0f;
    const float v[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, b.x * sc, b.y * sc, b.z * sc, b.w * sc};
    half8 hi, lo;
    unsigned vm = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        vm = range_acc(vm, v[j]);
        hi[j] = (f16)v[j];
        lo[j] = (f16)((v[j] - (float)hi[j]) * 2048.0f);
    }
    range_raise(fault, vm);
    *(half8*)(out + row * 2 * C + c) = hi;
    *(half8*)(out + row * 2 * C + C + c) = lo;
}
// fp32 [cout][taps][cin] -> f16 [cout][taps][3 cin]: [wh * 2^11 | wh | wl'] per tap (wl' = f16((w - wh) * 2^11))
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, long long total, int cin, int terms,
                                                            f16* __restrict__ out, const float* __restrict__ scale) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= total) return;
    const long long rt = i / cin;            // (cout, tap)
    const int c = (int)(i - rt * cin);
    const float v = scale ? w[i] * scale[0] : w[i];
    const f16 wh = (f16)v;
    f16* o = out + rt * terms * cin;
    o[c] = (f16)((float)wh * 2048.0f);
    o[cin + c] = wh;
    o[2 * cin + c] = (f16)((v - (float)wh) * 2048.0f);
    if (terms == 4) o[3 * cin + c] = (f16)(v - (float)wh);   // xl'.wl / 2^11 on the accumulator's scale: the unscaled low part
}
}  // namespace

int launch_split_pack(reid_ctx* ctx, const float* x, long long rows, int C, _Float16* out, const float* d_scale) {
    ARG_CHECK(C % 8 == 0);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)rows * C * 8.0);
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((rows * (C / 8) + 255) / 256)), dim3(256), 0, ctx->stream, x, rows, C, out,
                       ctx->fault, d_scale);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_split_weights(reid_ctx* ctx, const float* w, int cout, int taps, int cin, int terms, _Float16* out, const float* d_scale) {
    const long long total = (long long)cout * taps * cin;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, w, total, cin, terms, out, d_scale);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_conv3x3_split(reid_ctx* ctx, const Gemm16Params& p, int kind, double flops, double bytes) {
    ARG_CHECK(conv3x3_f16_supported(p) && (p.split_terms == 3 || p.split_terms == 4) && p.Cin % (64 * p.split_terms) == 0 && p.zero_page &&
              p.ldb % 8 == 0 && p.M % 128 == 0 && p.C32);
    prof_begin(ctx, kind, flops, bytes);
    int st;
    if (conv3x3_x3_supported(ctx, p)) {          // large launches: two independent 4-wave blocks per CU (conv3x3_x3.hip)
        st = launch_conv3x3_x3(ctx, p);
        prof_end(ctx);
        return st;
    }
    if (ctx->f16_loader_waves) {
        if (p.W == 32) st = launch_geom<32, 1, 1, true>(ctx, p);
        else if (p.W == 16) st = launch_geom<16, 1, 1, true>(ctx, p);
        else st = launch_geom<8, 2, 1, true>(ctx, p);
    } else {
#ifdef REID_EXPERIMENTS       // the builds without loader waves (switch f16_loader_waves = 0: 0.806 against 0.856 of peak): experiment builds only
        if (p.W == 32) st = launch_geom<32, 1, 0, true>(ctx, p);
        else if (p.W == 16) st = launch_geom<16, 1, 0, true>(ctx, p);
        else st = launch_geom<8, 2, 0, true>(ctx, p);
#else
        reid_set_error("conv3x3_f16: the builds without loader waves need a library made with -DREID_EXPERIMENTS");
        st = REID_ERR_ARG;
#endif
    }
    prof_end(ctx);
    return st;
}
