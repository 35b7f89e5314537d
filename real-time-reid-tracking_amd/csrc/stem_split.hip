// Stem of the fp32-class mode (precision 2): conv 7x7 stride 2 pad 3 (3 -> 64) + folded BatchNorm, no ReLU, + MaxPool2d(3, 2, 1)
// (SERes18_IBN.py:251-254) on v_mfma_f32_32x32x16_f16 with split operands, three f16 products per multiply, fp32 accumulators:
//   * fp32 input: x = xh + xl'/2^11, w = wh + wl'/2^11, x w = xh wh + (xl' wh + xh wl') / 2^11 (main + correction accumulator);
//   * uint8 crops: the crop preprocessing of the DeepSORT extractor (feature_extractor.py:41-46: x / 255 -> Normalize(0.5, 0.5))
//     is x = u / 255 with the INTEGER u = 2 v - 255, which one f16 holds exactly, and zero padding stays zero: the convolution
//     runs on u with no low part, the weight gets a THIRD part instead (w = wh + wl'/2^11 + wll'/2^22: u w = u wh + u wl'/2^11 +
//     u wll'/2^22, exact products, three accumulators) and the 1/255 goes into the BatchNorm scale.  u / 255 is the value the
//     reference's fp32 expression ((v / 255) - 0.5) / 0.5 rounds to within 2^-25 of |x| <= 1.  With the two-part weight the
//     config-1 noise vectors had their one 1-ulp row flip (tools/check_split_mode.py); with three parts 0 of 256 on both sets.
// The exact-fp32 stem (stem_f32.hip) spends 168 64-cycle MFMAs on a 32-pixel x 64-channel tile; this one 84 32-cycle MFMAs.
//
// Same walk as stem_f32.hip: a block owns a strip of one image and goes down it two output rows (128 pixels) at a time; the MaxPool
// runs on the accumulators.  What differs:
//   * input pixels live in LDS with FOUR channels (r g b 0) of f16 (fp32 input: one image for xh and one for xl'): a pixel is 8
//     bytes and an output column's 7-tap window starts on a 16-byte boundary (stride 2), so the 8 K-values of a lane are one
//     ds_read_b128 = two pixels.  A kernel row is 8 pixels (7 taps + one with zero weights) = K 32 = two MFMA steps; 14 per tile.
//   * the input rows are a ring of 32 LDS rows indexed by (iy - 2) & 31; a tile needs nine, four of them new: each of the 512
//     threads fetches ONE pixel of the group of four rows a tile commits (for the tile two ahead) one tile earlier.
//   * 8 waves: wave w computes columns 16 (w & 3) .. + 15 of both output rows for channels 32 (w >> 2) .. + 31; waves 4-7 run a
//     tile's epilogue one tile late, beside the MFMAs of their SIMD's other wave.
//   * the epilogue also writes the block input of layer 1 as [xh | xl'] (what split_pack_kernel would make of the fp32 output).
// Measured (1024 crops, tools/profile_split_pass.py): 0.75 ms against 1.46 + 0.10 (stem_f32 + split_pack); by ablation the MFMAs
// are 0.30 of it, the stores 0.16, the epilogue's vector instructions ~0.15, the fetch 0.06, block prologues the rest - additive,
// i.e. the phases of a tile do not overlap much even with the late epilogue; 30 crops: 42 us against 71.
#include "reid_internal.h"

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int IMG_H = 256, IMG_W = 128, OUT_H = 128, OUT_W = 64;
constexpr int PITCH = 144 * 8;      // bytes per LDS input row: 3 zero pixels + 128 + 5 (zero pixels; the 8th tap of column 63 reads pixel 133), padded so
                                    // that two rows are a multiple of 256 B: the two output rows of a ds_read_b128 lane group then hit disjoint banks
constexpr int RING = 32;            // LDS input rows
constexpr int WROW = 14 * 32 + 16;  // bytes per output channel of a weight image: 14 K-steps x 2 halves x 8 f16, + 16 (bank spread)

__device__ __forceinline__ unsigned split_pair(float x) {     // low 16 bits: xh, high 16 bits: xl' = f16((x - xh) 2^11)
    const f16 h = (f16)x;
    const f16 l = (f16)((x - (float)h) * 2048.0f);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}

template <bool U8>
__global__ __launch_bounds__(512, 1) void stem_split_kernel(const void* __restrict__ x, const float* __restrict__ wgt,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            int nseg, float* __restrict__ out, f16* __restrict__ packed,
                                                            int* __restrict__ fault) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const img_h = smem;                            // [RING][144] pixels of 4 f16
    unsigned char* const img_l = img_h + RING * PITCH;
    unsigned char* const w_h = img_l + RING * PITCH;              // [64][WROW]
    unsigned char* const w_l = w_h + 64 * WROW;
    float* const edge = (float*)(w_l + 64 * WROW);                // [2][4][64]: column 15 of every column group's vertical maxima, by tile parity

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int cw = wave & 3, cb = wave >> 2;
    // strips of an image: nseg of them, tiles [seg 64 / nseg, (seg + 1) 64 / nseg) - any nseg in 1 .. 64 (round 6: the launcher sizes the
    // strips so that the blocks fill ONE round of the chip where they can)
    const int img = blockIdx.x / nseg, seg = blockIdx.x - img * nseg;
    const int t0 = seg * (OUT_H / 2) / nseg, t1 = (seg + 1) * (OUT_H / 2) / nseg;

    for (int i = tid; i < 2 * RING * PITCH / 16; i += 512) ((uint4*)smem)[i] = uint4{0u, 0u, 0u, 0u};   // borders stay zero for good
    __syncthreads();
    // weights: stem.w is [64][8][24] fp32 (kernel row, 7 taps x 3 channels + 3 zeros) -> per channel [step s][half h][8]:
    // kernel row s >> 1, pixels 2 (2 (s & 1) + h) and + 1 of the row's 8, 4 channels each
    for (int i = tid; i < 64 * 14 * 16; i += 512) {
        const int n = i / 224, k = i - n * 224, s = k >> 4, h = (k >> 3) & 1, j = k & 7;
        const int px = 2 * (2 * (s & 1) + h) + (j >> 2), c = j & 3;
        const float v = (px < 7 && c < 3) ? wgt[n * 192 + (s >> 1) * 24 + px * 3 + c] : 0.f;
        const unsigned hl = split_pair(v);
        *(unsigned short*)(w_h + n * WROW + k * 2) = (unsigned short)hl;
        *(unsigned short*)(w_l + n * WROW + k * 2) = (unsigned short)(hl >> 16);
        if constexpr (U8) {      // third part of the weight (the xl' image is not used): w = wh + wl' / 2^11 + wll' / 2^22
            const float wh = (float)__builtin_bit_cast(f16, (unsigned short)hl), wl = (float)__builtin_bit_cast(f16, (unsigned short)(hl >> 16));
            const f16 ll = (f16)(((v - wh) * 2048.0f - wl) * 2048.0f);
            *(unsigned short*)(img_l + n * WROW + k * 2) = __builtin_bit_cast(unsigned short, ll);
        }
    }
    __syncthreads();

    // ---- staging: group j = input rows 4 j + 2 .. 4 j + 5, one pixel per thread
    const int f_row = tid >> 7, f_px = tid & 127;
    // uint8: the pixel's 3 bytes sit in the 8 bytes from the 4-byte boundary below them (the second dword clamped into the row:
    // byte 3 px & 3 is 0 or 1 for the last pixels, which then need the first dword only); fp32: its 12 bytes
    const int f_off = (3 * f_px) & ~3, f_off2 = f_off + 4 < IMG_W * 3 ? f_off + 4 : f_off, f_sh = 8 * ((3 * f_px) & 3);
    auto fetch = [&](int j, unsigned (&raw)[3]) __attribute__((always_inline)) {
        const int iy = 4 * j + 2 + f_row;
        raw[0] = raw[1] = raw[2] = 0u;
        if ((unsigned)iy < (unsigned)IMG_H) {
            if constexpr (U8) {
                const uint8_t* p = (const uint8_t*)x + ((long long)img * IMG_H + iy) * (IMG_W * 3);
                raw[0] = *(const unsigned*)(p + f_off);
                raw[1] = *(const unsigned*)(p + f_off2);
            } else {
                const float* p = (const float*)x + (((long long)img * IMG_H + iy) * IMG_W + f_px) * 3;
                raw[0] = __float_as_uint(p[0]); raw[1] = __float_as_uint(p[1]); raw[2] = __float_as_uint(p[2]);
            }
        }
    };
    auto commit = [&](int j, const unsigned (&raw)[3]) __attribute__((always_inline)) {
        const int iy = 4 * j + 2 + f_row;
        unsigned a = 0u, b = 0u, c = 0u;
        const int at = ((iy - 2) & (RING - 1)) * PITCH + (3 + f_px) * 8;
        if constexpr (U8) {      // u = 2 v - 255: an integer of at most 9 bits, exact in f16; rows outside the image stay 0
            if ((unsigned)iy < (unsigned)IMG_H) {
                const unsigned px = (unsigned)((((unsigned long long)raw[1] << 32) | raw[0]) >> f_sh);
                a = __builtin_bit_cast(unsigned short, (f16)(float)(2 * (int)(px & 0xffu) - 255));
                b = __builtin_bit_cast(unsigned short, (f16)(float)(2 * (int)((px >> 8) & 0xffu) - 255));
                c = __builtin_bit_cast(unsigned short, (f16)(float)(2 * (int)((px >> 16) & 0xffu) - 255));
            }
            *(u32x2*)(img_h + at) = u32x2{a | (b << 16), c};
            return;
        }
        if ((unsigned)iy < (unsigned)IMG_H) {
            a = split_pair(__uint_as_float(raw[0])); b = split_pair(__uint_as_float(raw[1])); c = split_pair(__uint_as_float(raw[2]));
        }
        *(u32x2*)(img_h + at) = u32x2{(a & 0xffffu) | (b << 16), c & 0xffffu};
        *(u32x2*)(img_l + at) = u32x2{(a >> 16) | (b & 0xffff0000u), c >> 16};
    };

    // this lane's output pixel of a tile: row (li >> 4) of the pair, column 16 cw + (li & 15); K half lh
    const int ox = cw * 16 + (li & 15);
    const int a_col = (2 * ox + 2 * lh) * 8;                         // + 32 (s & 1): pixel pairs 2 (s & 1) + lh of the row's four
    const int b_at = (cb * 32 + li) * WROW + lh * 16;                // + 32 s
    const float cs = U8 ? scale[cb * 32 + li] / 255.0f : scale[cb * 32 + li], sh = shift[cb * 32 + li];

    // a strip that does not start at the top of the image first computes the tile above it, for its second row only
    const int t_first = t0 > 0 ? t0 - 1 : t0;
    const int ntile = t1 - t_first;
    float prev[8], pend = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) prev[e] = -INFINITY;                 // row -1 of the image: MaxPool2d pads with -inf
    float* const pool_img = out + (long long)img * (OUT_H / 2) * (OUT_W / 2) * 64;
    f16* const pack_img = packed ? packed + (long long)img * (OUT_H / 2) * (OUT_W / 2) * 128 : nullptr;
    unsigned vmag = 0u;   // range guard (reid_ctx.fault): largest magnitude packed by this lane
    auto store = [&](int t, int col, float v) __attribute__((always_inline)) {                      // pooled pixel (t, col), this lane's channel
        const long long pix = (long long)t * (OUT_W / 2) + col;
        pool_img[pix * 64 + cb * 32 + li] = v;
        if (pack_img) {
            vmag = range_acc(vmag, v);
            const unsigned hl = split_pair(v);
            *(unsigned short*)(pack_img + pix * 128 + cb * 32 + li) = (unsigned short)hl;
            *(unsigned short*)(pack_img + pix * 128 + 64 + cb * 32 + li) = (unsigned short)(hl >> 16);
        }
    };
    // pooled column 8 cw of tile t from its two local columns (pend) and column 15 of group cw - 1 (edge, written during tile t)
    auto finish_first_column = [&](int t) __attribute__((always_inline)) {
        if (cw > 0 && lh == 0) store(t, cw * 8, fmaxf(pend, edge[((t & 1) * 4 + cw - 1) * 64 + cb * 32 + li]));
    };

    // the group a tile commits (two tiles ahead) was fetched one tile earlier: its loads have a whole tile to land in
    unsigned raw_a[3], raw_b[3];
    for (int j = t_first - 2; j <= t_first + 1; ++j) {
        fetch(j, raw_a);
        commit(j, raw_a);
    }
    const int t_last = t_first + ntile - 1;
    fetch(t_first + 2, raw_a);            // (a group past a short strip is fetched and never committed)
    __syncthreads();
    // the MFMAs of a tile: v = folded-BatchNorm output of this lane's 16 pixels (:252-253, no ReLU)
    auto mma = [&](int t) __attribute__((always_inline)) -> f32x16 {
        f32x16 acc, cor, cor2;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = cor[e] = cor2[e] = 0.f;
        const int row0 = 4 * t - 5 + 2 * (li >> 4);                  // ring row of kernel row r: (row0 + r) & 31
        // operands of step s + 2 are read before the MFMAs of step s are issued (three register sets): while this wave runs its
        // MFMAs the SIMD's other wave is in its epilogue, so nothing else covers the LDS latency
        half8 av[3], alv[3], bhv[3], blv[3];
        auto load_step = [&](int st, int buf) __attribute__((always_inline)) {
            const int a_at = ((row0 + (st >> 1)) & (RING - 1)) * PITCH + a_col + 32 * (st & 1);
            av[buf] = *(const half8*)(img_h + a_at);
            bhv[buf] = *(const half8*)(w_h + b_at + 32 * st);
            blv[buf] = *(const half8*)(w_l + b_at + 32 * st);
            if constexpr (U8) alv[buf] = *(const half8*)(img_l + b_at + 32 * st);     // third part of the weights
            else alv[buf] = *(const half8*)(img_l + a_at);
        };
        load_step(0, 0);
        load_step(1, 1);
#pragma unroll
        for (int st = 0; st < 14; ++st) {
            const int buf = st % 3;
            if (st + 2 < 14) load_step(st + 2, (st + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);      // keep the reads above the MFMAs: the compiler otherwise reads right before each use
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[buf], bhv[buf], acc, 0, 0, 0);
            if constexpr (U8) cor2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[buf], alv[buf], cor2, 0, 0, 0);
            else cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(alv[buf], bhv[buf], cor, 0, 0, 0);
            cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[buf], blv[buf], cor, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr float S1 = 1.0f / 2048.0f;
        f32x16 v;
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = (acc[e] + (cor[e] + cor2[e] * S1) * S1) * cs + sh;
        return v;
    };
    // max-pool + stores of a tile.  C layout: col = lane & 31 (channel); row m = (e & 3) + 8 (e >> 2) + 4 lh = pixel (row e >> 3 of the
    // pair, column 16 cw + (e & 3) + 8 ((e >> 2) & 1) + 4 lh)
    auto epilogue = [&](int t, const f32x16 v) __attribute__((always_inline)) {
        const bool emit = t >= t0;
        if (emit && t > t_first && t > t0) finish_first_column(t - 1);     // edge[(t - 1) & 1] was completed before the last barrier
        float vm[8];      // vertical maxima of this lane's 8 columns: lh 0 -> 0 1 2 3 8 9 10 11, lh 1 -> 4 5 6 7 12 13 14 15
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            vm[e] = fmaxf(fmaxf(prev[e], v[e]), v[e + 8]);
            prev[e] = v[e + 8];
        }
        if (emit) {
            // pooled column p = max of columns 2p-1, 2p, 2p+1.  In-lane: p = 1 + 2 lh (vm 1 2 3) and 5 + 2 lh (vm 5 6 7);
            // across the halves: p = 2 (col 3 | 4 5), 4 (col 7 | 8 9), 6 (col 11 | 12 13)
            const float m01 = fmaxf(vm[0], vm[1]), m45 = fmaxf(vm[4], vm[5]);
            const float pa = fmaxf(fmaxf(vm[1], vm[2]), vm[3]), pb = fmaxf(fmaxf(vm[5], vm[6]), vm[7]);
            const float s0 = lh ? m01 : vm[3], s1 = lh ? vm[3] : m45, s2 = lh ? m45 : vm[7];
            const float p2 = fmaxf(s0, __shfl_xor(s0, 32)), p4 = fmaxf(s1, __shfl_xor(s1, 32)), p6 = fmaxf(s2, __shfl_xor(s2, 32));
            store(t, cw * 8 + 1 + 2 * lh, pa);
            store(t, cw * 8 + 5 + 2 * lh, pb);
            store(t, cw * 8 + (lh ? 4 : 2), lh ? p4 : p2);
            if (lh) {
                store(t, cw * 8 + 6, p6);
                edge[((t & 1) * 4 + cw) * 64 + cb * 32 + li] = vm[7];      // column 15: the next group's column -1
            } else if (cw == 0) {
                store(t, 0, m01);                                          // column -1 is the image border
            } else {
                pend = m01;
            }
        }
    };
    // The two waves of a SIMD (w and w + 4: the two channel halves of the same pixels) reach every barrier together; if both ran
    // MFMAs then epilogue, the matrix pipe would idle through both epilogues.  Waves 4-7 run the epilogue of tile t - 1 BEFORE the
    // MFMAs of tile t, so on every SIMD one wave's ~250 vector instructions sit beside the other's 42 MFMAs.
    const bool defer = cb == 1;
    f32x16 vkeep;
#pragma unroll
    for (int e = 0; e < 16; ++e) vkeep[e] = 0.f;
    auto tile = [&](int t, unsigned (&raw_commit)[3], unsigned (&raw_fetch)[3]) __attribute__((always_inline)) {
        if (t + 3 <= t_last) fetch(t + 3, raw_fetch);
        if (defer) {
            if (t > t_first) epilogue(t - 1, vkeep);
            vkeep = mma(t);
        } else {
            vkeep = mma(t);
        }
        // after the MFMAs and before this tile's stores: the wait for the pixels fetched a tile ago is a wait for every older
        // vector-memory operation, and the stores before it have had the MFMAs' time to drain
        if (t + 2 <= t_last) commit(t + 2, raw_commit);
        if (!defer) epilogue(t, vkeep);
        __syncthreads();
    };
    for (int t = t_first; t <= t_last; t += 2) {
        tile(t, raw_a, raw_b);
        if (t + 1 <= t_last) tile(t + 1, raw_b, raw_a);
    }
    if (defer) epilogue(t_last, vkeep);
    __syncthreads();
    finish_first_column(t1 - 1);
    range_raise(fault, vmag);
}

constexpr int SMEM = 2 * RING * PITCH + 2 * 64 * WROW + 2 * 4 * 64 * 4;

}  // namespace

// x: uint8 NHWC crops (is_u8) or fp32 NHWC, both [n][256][128][3]; wgt: [64][8][24] fp32 (stem.w);
// out: [n][64][32][64] fp32 = conv + BN + MaxPool(3,2,1); packed (may be null): the same as [n][64][32][xh 64 | xl' 64] f16
int launch_stem_split(reid_ctx* ctx, const void* x, bool is_u8, int n, const float* wgt, const float* scale, const float* shift, float* out,
                      _Float16* packed) {
    ARG_CHECK(n >= 1);
    // A block walks one strip of one image (and redoes the tile above its strip), one block per CU.  Cost of a launch in fifths of a
    // tile time (measured at 20-33 crops: a tile ~4.4 us when the chip is this empty, a block's prologue ~0.6 of one):
    // rounds x (3 + 5 (strip length [+ 1 redone])).  Strips of ANY length (round 6; until then 64 / 2^k tiles): the strip count that
    // minimises that cost, i.e. for a tracking-sized batch the largest one whose blocks still fit one round - 33 crops: 7 strips = 231
    // blocks in one round instead of 16 = 528 in three (73.7 -> 46.8 us); 30 crops: 8 strips (42.5 us) cost what 16 did (42.0).
    // The tiles are the same tiles: bit-identical output (tools/probes/embed_hash.py against the previous build, 12 pass sizes).
    int nseg = 1;
    if (n < 256) {
        long long best = -1;
        for (int s = 1; s <= 64; ++s) {
            const long long rounds = ((long long)n * s + 255) / 256;
            const long long cost = rounds * (3 + 5 * ((64 + s - 1) / s + (s > 1 ? 1 : 0)));
            if (best < 0 || cost < best) { best = cost; nseg = s; }
        }
    }
    const int grid = n * nseg;
    // per device, so not cached in a static: contexts of one process may sit on different GPUs
    if (is_u8) HIP_TRY(hipFuncSetAttribute((const void*)stem_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    else HIP_TRY(hipFuncSetAttribute((const void*)stem_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    const double flops = 2.0 * n * OUT_H * OUT_W * 64 * 147.0;
    const double bytes = (double)n * IMG_H * IMG_W * 3 * (is_u8 ? 1.0 : 4.0) + (double)n * OUT_H * OUT_W * 64 * (packed ? 2.0 : 1.0) + 64 * 147 * 4.0;
    prof_begin(ctx, REID_K_CONV_GEMM, flops, bytes);
    if (is_u8) hipLaunchKernelGGL((stem_split_kernel<true>), dim3(grid), dim3(512), SMEM, ctx->stream, x, wgt, scale, shift, nseg, out, packed,
                                  ctx->fault);
    else hipLaunchKernelGGL((stem_split_kernel<false>), dim3(grid), dim3(512), SMEM, ctx->stream, x, wgt, scale, shift, nseg, out, packed, ctx->fault);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
