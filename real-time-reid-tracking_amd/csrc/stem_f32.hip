// Stem of the exact-fp32 path: conv 7x7 stride 2 pad 3 (3 -> 64) + folded BatchNorm, no ReLU (SERes18_IBN.py:251-253), on
// v_mfma_f32_32x32x2_f32, with the crop preprocessing of the DeepSORT extractor fused into the loader when the input is the
// uint8 crop itself (feature_extractor.py:41-46: x / 255 -> Normalize(0.5, 0.5)).
//
// Round 1 ran this as the generic implicit GEMM with a per-element predicated gather (K padded 147 -> 192): 47 TF/s, 10 % of the
// fp32 forward for 4 % of its FLOPs.  Here a block owns a strip of one image and walks it two output rows (128 pixels) at a time:
//   * the 64 x 7 x 24 weights (21 taps of a kernel row + 3 zeros; K = 168) sit in LDS for the whole strip,
//   * the nine input rows a tile needs are converted once into an fp32 LDS image with zero borders; the A operand of the MFMA is
//     read straight from it - for kernel row r the 21 taps of an output pixel are 21 CONSECUTIVE floats (NHWC, 3 channels),
//     so no im2col expansion exists anywhere,
//   * lane half h reads taps 8g + 4h .. + 3 of a row (two ds_read_b64) and the weights in the same order (one ds_read_b128):
//     each of the four MFMAs of a group sums k in {8g + e, 8g + 4 + e}, the same pairing on both operands,
//   * the next tile's rows are fetched before the 168 MFMAs of this tile are issued and written to the other LDS buffer after
//     them; one barrier per tile.
// A wave owns both output rows of a tile for 16 columns (MFMA row i -> row i >> 4, column 16 w + (i & 15)), so that the
// MaxPool2d(3, 2, 1) that follows (SERes18_IBN.py:254) can run on the accumulators (POOL): the vertical maximum of rows 2t-1
// (kept from the previous tile), 2t, 2t+1 is register-local (C rows e and e + 8), the horizontal one needs the other lane half
// (three values swapped across lanes l and l ^ 32) and, for the first pooled column of a wave, the last column of the wave to
// its left (256 B through LDS, consumed one tile later so the tile's one barrier covers it).  The 2 MiB / crop conv map then
// never reaches HBM (it cost 0.6 ms of maxpool kernel and 2 GB of writes per 1024 crops).
// Output: POOL [n][64][32][64], else [n][128][64][64] fp32 NHWC.
#include "reid_internal.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int IMG_H = 256, IMG_W = 128, OUT_H = 128, OUT_W = 64;
constexpr int PITCH = 408;     // floats per LDS input row: 9 (3 zero pixels) + 384 + 15 (zero pixels and slack for the padded taps)
constexpr int WP = 172;        // floats per LDS weight row: 7 x 24 + 4 (ds_read_b128 of 16 different rows: conflict-free)
constexpr int ROWS = 9;        // input rows of a two-row output tile

template <bool U8, bool POOL>
__global__ __launch_bounds__(256, 2) void stem_f32_kernel(const void* __restrict__ x, const float* __restrict__ wgt,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          int tiles_per_block, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float w_lds[64 * WP];
    __shared__ __attribute__((aligned(16))) float in_lds[2][ROWS * PITCH];
    __shared__ float edge[2][4][64];     // POOL: column 15 of every wave's vertical maxima, by tile parity
    __shared__ float norm_lut[256];      // U8: (v / 255 - 0.5) / 0.5 of every byte value - the two IEEE divisions once per block,
                                         // not ~25 VALU instructions per pixel value beside the MFMAs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int nseg = (OUT_H / 2) / tiles_per_block;
    const int img = blockIdx.x / nseg, seg = blockIdx.x - img * nseg;
    const int t0 = seg * tiles_per_block;

    for (int i = tid; i < 2 * ROWS * PITCH; i += 256) (&in_lds[0][0])[i] = 0.f;      // borders and slack stay zero for good
    for (int i = tid; i < 64 * 168; i += 256) {                                        // weights [64][8][24] -> rows 0..6
        const int nrow = i / 168, k = i - nrow * 168;
        w_lds[nrow * WP + k] = wgt[nrow * 192 + k];
    }
    for (int i = tid; i < 64 * 4; i += 256) w_lds[(i >> 2) * WP + 168 + (i & 3)] = 0.f;
    if constexpr (U8) norm_lut[tid] = ((float)tid / 255.0f - 0.5f) / 0.5f;   // feature_extractor.py:41-46, same arithmetic
    __syncthreads();

    // ---- staging: 16 consecutive channel values of one input row per thread and step
    // uint8: 9 rows x 24 pieces of 16 bytes (216 threads, one step); fp32: 9 rows x 96 pieces of 16 bytes (864 pieces, 4 steps)
    constexpr int STEPS = U8 ? 1 : 4;
    constexpr int PIECES = ROWS * (U8 ? 24 : 96);
    u32x4 raw[STEPS];
    auto fetch = [&](int t) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int p = tid + s * 256;
            const int per_row = U8 ? 24 : 96;
            const int row = p / per_row, piece = p - row * per_row;
            const int iy = 4 * t - 3 + row;
            raw[s] = u32x4{0u, 0u, 0u, 0u};
            if (p < PIECES && (unsigned)iy < (unsigned)IMG_H) {
                if constexpr (U8) raw[s] = *(const u32x4*)((const uint8_t*)x + ((long long)img * IMG_H + iy) * (IMG_W * 3) + piece * 16);
                else raw[s] = *(const u32x4*)((const float*)x + ((long long)img * IMG_H + iy) * (IMG_W * 3) + piece * 4);
            }
        }
    };
    auto commit = [&](int t, float* buf) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int p = tid + s * 256;
            const int per_row = U8 ? 24 : 96;
            const int row = p / per_row, piece = p - row * per_row;
            const int iy = 4 * t - 3 + row;
            if (p >= PIECES) continue;
            const bool inside = (unsigned)iy < (unsigned)IMG_H;
            if constexpr (U8) {
                float* dst = buf + row * PITCH + 9 + piece * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        dst[q * 4 + b] = inside ? norm_lut[(raw[s][q] >> (8 * b)) & 0xffu] : 0.f;
                    }
            } else {
                float* dst = buf + row * PITCH + 9 + piece * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = inside ? __uint_as_float(raw[s][q]) : 0.f;
            }
        }
    };

    // this wave's 32 output pixels of a tile: both output rows of the pair (li >> 4), columns wave * 16 + (li & 15)
    const int ox = wave * 16 + (li & 15);
    const int a_base = (2 * (li >> 4)) * PITCH + 6 * ox + 4 * lh;   // + r * PITCH + 8 g   (floats; even: ds_read_b64)
    const int b_base = li * WP + 4 * lh;                            // + 32 * WP * b + r * 24 + 8 g
    float cs[2], sh[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        cs[b] = scale[li + 32 * b];
        sh[b] = shift[li + 32 * b];
    }

    // POOL: a strip that does not start at the top of the image first computes the tile above it, for its second row only
    const int t_first = (POOL && t0 > 0) ? t0 - 1 : t0;
    const int ntile = t0 + tiles_per_block - t_first;
    float prev[2][8], pend[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        pend[b] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) prev[b][e] = -INFINITY;        // row -1 of the image: MaxPool2d pads with -inf
    }
    float* const pool_img = out + (long long)img * (OUT_H / 2) * (OUT_W / 2) * 64;
    // pooled column 8 w of tile t from its two local columns (pend) and column 15 of wave w - 1 (edge, written during tile t)
    auto finish_first_column = [&](int t) {
        if (wave > 0 && lh == 0) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
                pool_img[((long long)t * (OUT_W / 2) + wave * 8) * 64 + li + 32 * b] = fmaxf(pend[b], edge[t & 1][wave - 1][li + 32 * b]);
        }
    };

    fetch(t_first);
    commit(t_first, in_lds[0]);
    __syncthreads();
    for (int ti = 0; ti < ntile; ++ti) {
        const int t = t_first + ti;
        const float* cur = in_lds[ti & 1];
        const bool more = ti + 1 < ntile;
        if (more) fetch(t + 1);
        f32x16 acc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const f32x2 a0 = *(const f32x2*)(cur + a_base + r * PITCH + 8 * g);
                const f32x2 a1 = *(const f32x2*)(cur + a_base + r * PITCH + 8 * g + 2);
                const float av[4] = {a0.x, a0.y, a1.x, a1.y};
                f32x4 bv[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) bv[b] = *(const f32x4*)(w_lds + b_base + 32 * WP * b + r * 24 + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[b][e], acc[b], 0, 0, 0);
            }
        // epilogue: folded BatchNorm, no ReLU (:252-253).  C layout: col = lane & 31 (channel); row m = (e & 3) + 8 (e >> 2) + 4 lh
        // = pixel (row m >> 4 = e >> 3 of the pair, column 16 w + (e & 3) + 8 ((e >> 2) & 1) + 4 lh)
        if constexpr (!POOL) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int oy = 2 * t + (e >> 3), col = wave * 16 + (e & 3) + 8 * ((e >> 2) & 1) + 4 * lh;
                    out[(((long long)img * OUT_H + oy) * OUT_W + col) * 64 + li + 32 * b] = acc[b][e] * cs[b] + sh[b];
                }
        } else {
            const bool emit = t >= t0;
            if (emit && ti > 0 && t > t0) finish_first_column(t - 1);     // edge[(t-1) & 1] was completed by the previous barrier
            float* orow = pool_img + ((long long)t * (OUT_W / 2) + wave * 8) * 64 + li;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float vm[8];      // vertical maxima of this lane's 8 columns: lh 0 -> 0 1 2 3 8 9 10 11, lh 1 -> 4 5 6 7 12 13 14 15
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float top = acc[b][e] * cs[b] + sh[b], bot = acc[b][e + 8] * cs[b] + sh[b];
                    vm[e] = fmaxf(fmaxf(prev[b][e], top), bot);
                    prev[b][e] = bot;
                }
                if (emit) {
                    // pooled column p = max of columns 2p-1, 2p, 2p+1.  In-lane: p = 1 + 2 lh (vm 1 2 3) and 5 + 2 lh (vm 5 6 7);
                    // across the halves: p = 2 (col 3 | 4 5), 4 (col 7 | 8 9), 6 (col 11 | 12 13)
                    const float m01 = fmaxf(vm[0], vm[1]), m45 = fmaxf(vm[4], vm[5]);
                    const float pa = fmaxf(fmaxf(vm[1], vm[2]), vm[3]), pb = fmaxf(fmaxf(vm[5], vm[6]), vm[7]);
                    const float s0 = lh ? m01 : vm[3], s1 = lh ? vm[3] : m45, s2 = lh ? m45 : vm[7];
                    const float p2 = fmaxf(s0, __shfl_xor(s0, 32)), p4 = fmaxf(s1, __shfl_xor(s1, 32)), p6 = fmaxf(s2, __shfl_xor(s2, 32));
                    orow[(1 + 2 * lh) * 64 + 32 * b] = pa;
                    orow[(5 + 2 * lh) * 64 + 32 * b] = pb;
                    orow[(lh ? 4 : 2) * 64 + 32 * b] = lh ? p4 : p2;
                    if (lh) {
                        orow[6 * 64 + 32 * b] = p6;
                        edge[t & 1][wave][li + 32 * b] = vm[7];          // column 15: the next wave's column -1
                    } else if (wave == 0) {
                        orow[32 * b] = m01;                              // column -1 is the image border
                    } else {
                        pend[b] = m01;
                    }
                }
            }
        }
        if (more) commit(t + 1, in_lds[(ti + 1) & 1]);
        __syncthreads();
    }
    if constexpr (POOL) finish_first_column(t0 + tiles_per_block - 1);
}

}  // namespace

// x: uint8 NHWC crops (is_u8) or fp32 NHWC, both [n][256][128][3]; wgt: [64][8][24] (stem.w);
// out: pooled ? [n][64][32][64] (conv + BN + MaxPool(3,2,1)) : [n][128][64][64]
int launch_stem_f32(reid_ctx* ctx, const void* x, bool is_u8, int n, const float* wgt, const float* scale, const float* shift, float* out,
                    bool pooled) {
    ARG_CHECK(n >= 1);
    // a block walks tiles_per_block tiles of one image; enough blocks to fill the chip twice over when there are few images
    int tpb = 64;
    while (tpb > (pooled ? 8 : 1) && (long long)n * (64 / tpb) < 512) tpb >>= 1;   // pooled strips redo one tile each
    const int grid = n * (64 / tpb);
    const double flops = 2.0 * n * OUT_H * OUT_W * 64 * 147.0;
    const double bytes = (double)n * IMG_H * IMG_W * 3 * (is_u8 ? 1.0 : 4.0) + (double)n * OUT_H * OUT_W * 64 * (pooled ? 1.0 : 4.0) + 64 * 147 * 4.0;
    prof_begin(ctx, REID_K_CONV_GEMM, flops, bytes);
    if (pooled) {
        if (is_u8) hipLaunchKernelGGL((stem_f32_kernel<true, true>), dim3(grid), dim3(256), 0, ctx->stream, x, wgt, scale, shift, tpb, out);
        else hipLaunchKernelGGL((stem_f32_kernel<false, true>), dim3(grid), dim3(256), 0, ctx->stream, x, wgt, scale, shift, tpb, out);
    } else {
        if (is_u8) hipLaunchKernelGGL((stem_f32_kernel<true, false>), dim3(grid), dim3(256), 0, ctx->stream, x, wgt, scale, shift, tpb, out);
        else hipLaunchKernelGGL((stem_f32_kernel<false, false>), dim3(grid), dim3(256), 0, ctx->stream, x, wgt, scale, shift, tpb, out);
    }
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
