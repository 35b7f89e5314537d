// Stem of the fp16 embed path as ONE kernel: conv 7x7 / stride 2 / pad 3 (3 -> 64) + folded BN (no ReLU: the reference's
// forward never applies one, SERes18_IBN.py:250-254) + MaxPool2d(3, 2, 1).
//
// Why: unfused, the stem GEMM writes its 128x64x64 map (1 MiB f16 per crop, 537 MB per 512 crops) and the pool reads it
// back - 0.45 ms of a 3.6 ms forward spent on a tensor nobody else needs.  Fused, HBM sees the padded input once
// (285 KB/crop) and the pooled map once (256 KB/crop).
//
// One block (8 waves) walks one crop top to bottom, four conv rows (256 pixels) per step:
//   * the 64 x 224 weight matrix (BN scale folded in) lives in REGISTERS for the whole crop: as the MFMA A operand
//     (rows = output channels), 2 x 14 fragments = 112 VGPRs per lane - one block per CU, 256 VGPRs per wave;
//   * input rows sit in an LDS ring (24 padded NHWC4 rows of 1088 B); the B operand of tap-row r / k-step j for output
//     pixel x is the contiguous 16 bytes at row 2y+r, pixel 2x+4j+2*(lane>>5): consecutive lanes read consecutive
//     16-byte slots, so every ds_read_b128 is conflict-free and each fragment feeds two MFMAs;
//   * with channels on the accumulator rows a lane holds 4 consecutive channels of one pixel per register quad: the conv
//     tile goes to LDS as packed 8-byte stores (8 per lane per step), into an 8-row ring so the row shared by two pooling
//     windows is never copied;
//   * pooling reads 16-byte channel groups from that ring, adds the BN shift (max commutes with a per-channel shift) and
//     stores 16 bytes per lane, fully coalesced.
// Two block barriers per step; the next step's eight input rows are fetched into registers before the MFMA phase and
// written to the ring after it.
#include "reid_internal.h"

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef f16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int IN_H = 256, IN_W = 128;          // crop
constexpr int PH = 262, PW = 136;              // padded NHWC4 input (api.hip PAD_H / PAD_W)
constexpr int ROW_B = PW * 8;                  // 1088 bytes per padded input row
constexpr int RING_ROWS = 24;
constexpr int CH = 128, CW = 64;               // conv map
constexpr int OH = 64, OW = 32;                // pooled map
constexpr int TILE_ROW_B = CW * 128;           // one conv row: 64 pixels x 64 channels f16
constexpr int RING_BYTES = RING_ROWS * ROW_B;  // 26 112
constexpr int TILE_BYTES = 8 * TILE_ROW_B;     // 65 536

__device__ __forceinline__ int ring_slot(int padded_row) { return (padded_row + 3) % RING_ROWS; }

// U8: the input is the raw uint8 NHWC crop [256][128][3]; (v/255 - 0.5)/0.5 (feature_extractor.py:41-46), the channel
// padding to 4 and the zero border are produced while the ring is filled - the padded f16 image (285 KB/crop written and
// read back by a separate kernel) never exists.  Otherwise pad_in is that padded NHWC4 f16 image [262][136][4].
template <bool U8>
__global__ __launch_bounds__(512) void stem_pool_f16_kernel(const f16* __restrict__ pad_in, const uint8_t* __restrict__ crops,
                                                            const f16* __restrict__ w16s, const float* __restrict__ shift,
                                                            f16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char lds[RING_BYTES + TILE_BYTES];
    char* ring = lds;
    char* tile = lds + RING_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const char* img = U8 ? nullptr : (const char*)(pad_in + (long long)blockIdx.x * PH * PW * 4);
    const uint8_t* cimg = U8 ? crops + (long long)blockIdx.x * IN_H * IN_W * 3 : nullptr;
    // padded pixel idx (row-major over [rows][136]) of padded row r0 + idx/136 -> the three bytes of the source pixel, or -1
    auto src_of = [&](int r0, int idx) -> int {
        const int r = r0 + idx / PW, pp = idx % PW;
        const int y = r - 3, x = pp - 3;
        return ((unsigned)y < (unsigned)IN_H && (unsigned)x < (unsigned)IN_W) ? (y * IN_W + x) * 3 : -1;
    };
    auto to_px = [&](int b0, int b1, int b2, bool ok) -> half4 {
        half4 v = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        if (ok) {
            v[0] = (f16)(((float)b0 / 255.0f - 0.5f) / 0.5f);
            v[1] = (f16)(((float)b1 / 255.0f - 0.5f) / 0.5f);
            v[2] = (f16)(((float)b2 / 255.0f - 0.5f) / 0.5f);
        }
        return v;
    };
    f16* o_img = out + (long long)blockIdx.x * OH * OW * 64;

    // weights -> registers: fragment (ct, ks) = channels ct*32 + li, k = (ks>>1)*32 + (ks&1)*16 + lh*8 .. +8
    half8 wf[2][14];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 14; ++ks)
            wf[ct][ks] = *(const half8*)(w16s + (ct * 32 + li) * 256 + (ks >> 1) * 32 + (ks & 1) * 16 + lh * 8);

    // ring: padded rows 0..12 (slots 3..15, contiguous); conv-tile slot of row -1: -inf for the first pooling window
    if constexpr (U8) {
        for (int idx = tid; idx < 13 * PW; idx += 512) {
            const int so = src_of(0, idx);
            const bool ok = so >= 0;
            const int o = ok ? so : 0;
            *(half4*)(ring + 3 * ROW_B + idx * 8) = to_px(cimg[o], cimg[o + 1], cimg[o + 2], ok);
        }
    } else {
        for (int idx = tid; idx < 13 * ROW_B / 16; idx += 512)
            *(half8*)(ring + 3 * ROW_B + idx * 16) = *(const half8*)(img + idx * 16);
    }
    {
        half8 ninf;
#pragma unroll
        for (int e = 0; e < 8; ++e) ninf[e] = (f16)(-65504.f);
        *(half8*)(tile + 7 * TILE_ROW_B + tid * 16) = ninf;
    }
    // pooling role of this thread: pooled row (tid >> 8) of the step, pooled column ox, channel octet cg
    const int p_oy = tid >> 8, p_ox = (tid >> 3) & 31, p_cg = tid & 7;
    float sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[e] = shift[p_cg * 8 + e];
    // MFMA role: conv row y of the step, pixels x0 .. x0+31
    const int y = wave >> 1, x0 = (wave & 1) * 32;
    const int px = x0 + li;
    __syncthreads();

    for (int k = 0; k < CH / 4; ++k) {
        // next step's eight input rows (padded rows 8k+13 .. 8k+20, contiguous in global memory and in the ring)
        half8 pre0, pre1;
        int pb[3][3];           // U8: the bytes of this thread's (up to) three pixels of the next eight rows
        bool pok[3];
        const bool more = k + 1 < CH / 4;
        if (more) {
            if constexpr (U8) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int idx = tid + 512 * i;
                    const int so = idx < 8 * PW ? src_of(8 * k + 13, idx) : -1;
                    pok[i] = so >= 0;
                    const int o = pok[i] ? so : 0;
                    pb[i][0] = cimg[o]; pb[i][1] = cimg[o + 1]; pb[i][2] = cimg[o + 2];
                }
            } else {
                const char* src = img + (long long)(8 * k + 13) * ROW_B;
                pre0 = *(const half8*)(src + tid * 16);
                if (tid < 8 * ROW_B / 16 - 512) pre1 = *(const half8*)(src + (512 + tid) * 16);
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
        const int row0 = 8 * k + 2 * y;   // padded input row of tap-row 0
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            const int r = ks >> 1, j = ks & 1;
            const half8 bf = *(const half8*)(ring + ring_slot(row0 + r) * ROW_B + (2 * px + 4 * j + 2 * lh) * 8);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][ks], bf, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][ks], bf, acc[1], 0, 0, 0);
        }
        __syncthreads();   // everyone is done with the ring rows of this step and with the pooling of the previous one
        {
            char* trow = tile + ((4 * k + y) & 7) * TILE_ROW_B + px * 128;
            const int key = (px >> 1) & 7;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (f16)acc[ct][q * 4 + e];
                    *(half4*)(trow + (((ct * 4 + q) ^ key) * 16) + lh * 8) = v;
                }
        }
        if (more) {
            char* dst = ring + ring_slot(8 * k + 13) * ROW_B;
            if constexpr (U8) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int idx = tid + 512 * i;
                    if (idx < 8 * PW) *(half4*)(dst + idx * 8) = to_px(pb[i][0], pb[i][1], pb[i][2], pok[i]);
                }
            } else {
                *(half8*)(dst + tid * 16) = pre0;
                if (tid < 8 * ROW_B / 16 - 512) *(half8*)(dst + (512 + tid) * 16) = pre1;
            }
        }
        __syncthreads();
        // pooled row oy = 2k + p_oy: conv rows 2oy-1 .. 2oy+1 (row -1 holds -inf), columns 2ox-1 .. 2ox+1
        {
            const int oy = 2 * k + p_oy;
            half8 m;   // the maximum of f16 values is exact in f16: packed v_pk_max_f16, no conversions inside the window
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = (f16)(-65504.f);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const char* trow = tile + ((2 * oy - 1 + dy) & 7) * TILE_ROW_B;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int cx = 2 * p_ox - 1 + dx;
                    if (cx < 0) continue;
                    const half8 v = *(const half8*)(trow + cx * 128 + ((p_cg ^ ((cx >> 1) & 7)) * 16));
                    m = __builtin_elementwise_max(m, v);
                }
            }
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)((float)m[e] + sh[e]);
            *(half8*)(o_img + ((long long)oy * OW + p_ox) * 64 + p_cg * 8) = o;
        }
    }
}

// stem weights [64][192 = r*24 + s*3 + c] fp32 x BN scale -> f16 [64][256 = r*32 + s*4 + c], zero padded
__global__ void stem_w16_scaled_kernel(const float* __restrict__ w, const float* __restrict__ scale, f16* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 256) return;
    const int co = i >> 8, k = i & 255;
    const int r = k >> 5, s = (k >> 2) & 7, c = k & 3;
    float v = 0.f;
    if (r < 7 && s < 7 && c < 3) v = w[co * 192 + r * 24 + s * 3 + c] * scale[co];
    out[i] = (f16)v;
}

}  // namespace

int launch_stem_w16_scaled(reid_ctx* ctx, const float* w, const float* scale, f16* out) {
    hipLaunchKernelGGL(stem_w16_scaled_kernel, dim3(64), dim3(256), 0, ctx->stream, w, scale, out);
    LAUNCH_CHECK();
    return REID_OK;
}

// pad_in [n][262][136][4] f16, or crops_u8 [n][256][128][3] (then pad_in is not read) -> pooled [n][64][32][64] f16
int launch_stem_pool_f16(reid_ctx* ctx, const f16* pad_in, const uint8_t* crops_u8, int n, const f16* w16s, const float* shift,
                         f16* out) {
    const double flops = 2.0 * n * CH * CW * 64 * 147;
    const double bytes = (double)n * ((crops_u8 ? IN_H * IN_W * 3.0 : PH * PW * 8.0) + OH * OW * 128.0);
    prof_begin(ctx, REID_K_CONV_GEMM, flops, bytes);
    if (crops_u8) hipLaunchKernelGGL(stem_pool_f16_kernel<true>, dim3(n), dim3(512), 0, ctx->stream, pad_in, crops_u8, w16s, shift, out);
    else hipLaunchKernelGGL(stem_pool_f16_kernel<false>, dim3(n), dim3(512), 0, ctx->stream, pad_in, crops_u8, w16s, shift, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
