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
// Output: [n][128][64][64] fp32 NHWC (the maxpool kernel follows).
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

template <bool U8>
__global__ __launch_bounds__(256, 2) void stem_f32_kernel(const void* __restrict__ x, const float* __restrict__ wgt,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          int tiles_per_block, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float w_lds[64 * WP];
    __shared__ __attribute__((aligned(16))) float in_lds[2][ROWS * PITCH];
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
                        const float v = (float)((raw[s][q] >> (8 * b)) & 0xffu);
                        dst[q * 4 + b] = inside ? (v / 255.0f - 0.5f) / 0.5f : 0.f;   // feature_extractor.py:41-46
                    }
            } else {
                float* dst = buf + row * PITCH + 9 + piece * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = inside ? __uint_as_float(raw[s][q]) : 0.f;
            }
        }
    };

    // this wave's 32 output pixels of a tile: output row (wave >> 1) of the pair, columns (wave & 1) * 32 + li
    const int oyl = wave >> 1, ox = (wave & 1) * 32 + li;
    const int a_base = (2 * oyl) * PITCH + 6 * ox + 4 * lh;       // + r * PITCH + 8 g   (floats; even: ds_read_b64)
    const int b_base = li * WP + 4 * lh;                            // + 32 * WP * b + r * 24 + 8 g
    float cs[2], sh[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        cs[b] = scale[li + 32 * b];
        sh[b] = shift[li + 32 * b];
    }

    fetch(t0);
    commit(t0, in_lds[0]);
    __syncthreads();
    for (int ti = 0; ti < tiles_per_block; ++ti) {
        const int t = t0 + ti;
        const float* cur = in_lds[ti & 1];
        const bool more = ti + 1 < tiles_per_block;
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
        // epilogue: folded BatchNorm, no ReLU (:252-253); C layout col = lane & 31 (channel), rows = pixels
        const int oy = 2 * t + oyl;
        float* orow = out + (((long long)img * OUT_H + oy) * OUT_W + (wave & 1) * 32 + 4 * lh) * 64 + li;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) orow[((e & 3) + 8 * (e >> 2)) * 64 + 32 * b] = acc[b][e] * cs[b] + sh[b];
        if (more) commit(t + 1, in_lds[(ti + 1) & 1]);
        __syncthreads();
    }
}

}  // namespace

// x: uint8 NHWC crops (is_u8) or fp32 NHWC, both [n][256][128][3]; wgt: [64][8][24] (stem.w); out: [n][128][64][64]
int launch_stem_f32(reid_ctx* ctx, const void* x, bool is_u8, int n, const float* wgt, const float* scale, const float* shift, float* out) {
    ARG_CHECK(n >= 1);
    // a block walks tiles_per_block tiles of one image; enough blocks to fill the chip twice over when there are few images
    int tpb = 64;
    while (tpb > 1 && (long long)n * (64 / tpb) < 512) tpb >>= 1;
    const int grid = n * (64 / tpb);
    const double flops = 2.0 * n * OUT_H * OUT_W * 64 * 147.0;
    const double bytes = (double)n * IMG_H * IMG_W * 3 * (is_u8 ? 1.0 : 4.0) + (double)n * OUT_H * OUT_W * 64 * 4.0 + 64 * 147 * 4.0;
    prof_begin(ctx, REID_K_CONV_GEMM, flops, bytes);
    if (is_u8) hipLaunchKernelGGL(stem_f32_kernel<true>, dim3(grid), dim3(256), 0, ctx->stream, x, wgt, scale, shift, tpb, out);
    else hipLaunchKernelGGL(stem_f32_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream, x, wgt, scale, shift, tpb, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
