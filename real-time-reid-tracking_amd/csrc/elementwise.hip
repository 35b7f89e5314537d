// HBM-bound kernels of the embed path: preprocessing, max-pool, InstanceNorm/BatchNorm finalisation,
// squeeze-excite, SE combine, GeM + BNNeck, row norms.  All NHWC fp32, 16-byte accesses, wave64 shuffles.
#include "reid_internal.h"
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- NCHW fp32 [n][3][h][w] -> NHWC [n][h][w][3] (plugin surface hands over torch-layout batches)
__global__ void nchw_to_nhwc3_kernel(const float* __restrict__ x, long long npix, int hw, float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / hw;
        const int p = (int)(i - img * hw);
        const float* src = x + img * 3 * hw + p;
        float* dst = out + i * 3;
        dst[0] = src[0];
        dst[1] = src[hw];
        dst[2] = src[2 * hw];
    }
}

// ---- Extractor._preprocess (feature_extractor.py:31-46): u8/255 -> bilinear resize (cv2 INTER_LINEAR: half-pixel
// centres, edge clamp, no antialias, horizontal then vertical fp32 lerp) -> (x-0.5)/0.5, NHWC out.
// Unfused mul/add (_rn intrinsics) so the result is bit-identical to the numpy oracle.
__device__ __forceinline__ void lin_tap(int d, int dst, int src, int& s, float& f) {
    const double scale = (double)src / (double)dst;
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx = fx - (float)sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= src - 1) { sx = src - 1; fx = 0.f; }
    s = sx;
    f = fx;
}
// pitch: source row length in pixels (0: the crop's own width, packed crops; frame width when the crops are windows of a frame)
__global__ void resize_norm_kernel(const uint8_t* __restrict__ packed, const long long* __restrict__ offsets,
                                   const int* __restrict__ hw, int n, int H, int W, int pitch, float* __restrict__ out) {
    const long long total = (long long)n * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int img = (int)(i / (H * W));
        const int rem = (int)(i - (long long)img * H * W);
        const int dy = rem / W, dx = rem - dy * W;
        const int h = hw[2 * img], w = hw[2 * img + 1];
        const uint8_t* src = packed + offsets[img];
        const long long ps = pitch ? pitch : w;
        int sx, sy;
        float fx, fy;
        lin_tap(dx, W, w, sx, fx);
        lin_tap(dy, H, h, sy, fy);
        const int sx1 = min(sx + 1, w - 1), sy1 = min(sy + 1, h - 1);
        const float gx = __fsub_rn(1.0f, fx), gy = __fsub_rn(1.0f, fy);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p00 = (float)src[(sy * ps + sx) * 3 + c] / 255.0f;
            const float p01 = (float)src[(sy * ps + sx1) * 3 + c] / 255.0f;
            const float p10 = (float)src[(sy1 * ps + sx) * 3 + c] / 255.0f;
            const float p11 = (float)src[(sy1 * ps + sx1) * 3 + c] / 255.0f;
            const float r0 = __fadd_rn(__fmul_rn(p00, gx), __fmul_rn(p01, fx));
            const float r1 = __fadd_rn(__fmul_rn(p10, gx), __fmul_rn(p11, fx));
            const float v = __fadd_rn(__fmul_rn(r0, gy), __fmul_rn(r1, fy));
            out[i * 3 + c] = __fsub_rn(v, 0.5f) / 0.5f;
        }
    }
}

// ---- MaxPool2d(3, 2, 1) on NHWC, c % 4 == 0 (SERes18_IBN.py:254)
__global__ void maxpool3s2_kernel(const float* __restrict__ x, int n, int h, int w, int c, int ho, int wo,
                                  float* __restrict__ out) {
    const int c4n = c >> 2;
    const long long total = (long long)n * ho * wo * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c4n);
        long long t = i / c4n;
        const int ox = (int)(t % wo);
        t /= wo;
        const int oy = (int)(t % ho);
        const int img = (int)(t / ho);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)w) continue;
                const f32x4 v = *(const f32x4*)(x + (((long long)img * h + iy) * w + ix) * c + cc * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *(f32x4*)(out + (((long long)img * ho + oy) * wo + ox) * c + cc * 4) = m;
    }
}

// ---- IBN finalisation (SERes18_IBN.py:88-93): per-(image, channel) affine that the NEXT conv's loader applies.
// channels [0, half): InstanceNorm2d(affine) from the conv epilogue's per-tile sum / sumsq partials (biased variance,
// eps 1e-5); channels [half, c): eval BatchNorm folded on the host.  Partials are reduced in fp64 in a fixed order.
__global__ void norm_finalize_kernel(const float* __restrict__ stats, int tiles, int c, int half, int hw,
                                     const float* __restrict__ in_gamma, const float* __restrict__ in_beta,
                                     const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                     float* __restrict__ a_scale, float* __restrict__ a_shift) {
    const int img = blockIdx.x;
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        float a, b;
        if (ch < half) {
            double s1 = 0.0, s2 = 0.0;
            for (int t = 0; t < tiles; ++t) {
                const float* st = stats + (((long long)img * tiles + t) * c + ch) * 2;
                s1 += (double)st[0];
                s2 += (double)st[1];
            }
            const double mean = s1 / hw;
            double var = s2 / hw - mean * mean;
            if (var < 0.0) var = 0.0;
            const double inv = 1.0 / sqrt(var + 1e-5);
            a = (float)(inv * (double)in_gamma[ch]);
            b = (float)((double)in_beta[ch] - mean * inv * (double)in_gamma[ch]);
        } else {
            a = bn_scale[ch - half];
            b = bn_shift[ch - half];
        }
        a_scale[(long long)img * c + ch] = a;
        a_shift[(long long)img * c + ch] = b;
    }
}

// ---- InstanceNorm half of an IBN layer applied in place (SERes18_IBN.py:88-93), for the LDS-DMA convolution kernel
// (conv_f32.hip) whose loader cannot transform its input: x[img][pix][ch] = relu(x * a + b) for ch < half, with (a, b)
// from the producing conv's per-tile sum / sumsq partials exactly as norm_finalize_kernel computes them.  The BatchNorm
// half was finished in that conv's epilogue.  grid = (hw / rows, images): every block recomputes its image's (a, b).
__global__ __launch_bounds__(256) void in_apply_kernel(float* __restrict__ x, const float* __restrict__ stats, int tiles, int c,
                                                       int half, int hw, int rows, const float* __restrict__ in_gamma,
                                                       const float* __restrict__ in_beta) {
    __shared__ float sa[256], sb[256];
    const int img = blockIdx.y, tid = threadIdx.x;
    for (int ch = tid; ch < half; ch += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int t = 0; t < tiles; ++t) {
            const float* st = stats + (((long long)img * tiles + t) * c + ch) * 2;
            s1 += (double)st[0];
            s2 += (double)st[1];
        }
        const double mean = s1 / hw;
        double var = s2 / hw - mean * mean;
        if (var < 0.0) var = 0.0;
        const double inv = 1.0 / sqrt(var + 1e-5);
        sa[ch] = (float)(inv * (double)in_gamma[ch]);
        sb[ch] = (float)((double)in_beta[ch] - mean * inv * (double)in_gamma[ch]);
    }
    __syncthreads();
    const int q = half >> 2;   // 16-byte chunks of a pixel's InstanceNorm half
    float* base = x + ((long long)img * hw + (long long)blockIdx.x * rows) * c;
    for (int i = tid; i < rows * q; i += 256) {
        const int row = i / q, cc = i - row * q;
        f32x4* ptr = (f32x4*)(base + (long long)row * c + cc * 4);
        f32x4 v = *ptr;
        const f32x4 a = *(const f32x4*)&sa[cc * 4], b = *(const f32x4*)&sb[cc * 4];
        v = v * a + b;
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *ptr = v;
    }
}

// ---- precision 2: the same finish of an IBN layer, but the consumer (conv2 in fp32-class arithmetic) reads [xh | xl'] f16, so the
// result goes straight there: channels < half = relu(x * a + b), the BatchNorm half as the conv epilogue left it; nothing is
// written back in fp32 (conv2 is the only reader of this tensor).  One pass instead of in_apply + split_pack.
__global__ __launch_bounds__(256) void in_apply_pack_kernel(const float* __restrict__ x, const float* __restrict__ stats, int tiles, int c,
                                                            int half, int hw, int rows, const float* __restrict__ in_gamma,
                                                            const float* __restrict__ in_beta, _Float16* __restrict__ packed, int in_only,
                                                            int* __restrict__ fault) {
    __shared__ float sa[512], sb[512];
    const int img = blockIdx.y, tid = threadIdx.x;
    for (int ch = tid; ch < c; ch += 256) {
        if (ch >= half) {   // BatchNorm half: identity here
            sa[ch] = 1.f;
            sb[ch] = 0.f;
            continue;
        }
        double s1 = 0.0, s2 = 0.0;
        for (int t = 0; t < tiles; ++t) {
            const float* st = stats + (((long long)img * tiles + t) * c + ch) * 2;
            s1 += (double)st[0];
            s2 += (double)st[1];
        }
        const double mean = s1 / hw;
        double var = s2 / hw - mean * mean;
        if (var < 0.0) var = 0.0;
        const double inv = 1.0 / sqrt(var + 1e-5);
        sa[ch] = (float)(inv * (double)in_gamma[ch]);
        sb[ch] = (float)((double)in_beta[ch] - mean * inv * (double)in_gamma[ch]);
    }
    __syncthreads();
    // in_only: the BatchNorm half left the conv epilogue as [yh | yl'] already - only the InstanceNorm channels pass through here
    const int q = (in_only ? half : c) >> 2;
    const long long pix0 = (long long)img * hw + (long long)blockIdx.x * rows;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    unsigned vm = 0u;
    for (int i = tid; i < rows * q; i += 256) {
        const int row = i / q, cc = i - row * q;
        f32x4 v = *(const f32x4*)(x + (pix0 + row) * c + cc * 4);
        if (cc * 4 < half) {   // half % 4 == 0: a chunk is entirely InstanceNorm or entirely BatchNorm
            const f32x4 a = *(const f32x4*)&sa[cc * 4], b = *(const f32x4*)&sb[cc * 4];
            v = v * a + b;
            vm = range_acc(range_acc(range_acc(range_acc(vm, v.x), v.y), v.z), v.w);   // before the ReLU: max(NaN, 0) is 0
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else {
            vm = range_acc(range_acc(range_acc(range_acc(vm, v.x), v.y), v.z), v.w);
        }
        const h4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        const h4 lo = {(_Float16)((v.x - (float)hi.x) * 2048.0f), (_Float16)((v.y - (float)hi.y) * 2048.0f),
                       (_Float16)((v.z - (float)hi.z) * 2048.0f), (_Float16)((v.w - (float)hi.w) * 2048.0f)};
        *(h4*)(packed + (pix0 + row) * 2 * c + cc * 4) = hi;
        *(h4*)(packed + (pix0 + row) * 2 * c + c + cc * 4) = lo;
    }
    range_raise(fault, vm);
}

// ---- SEBlock (SERes18_IBN.py:32-41): s = sigmoid(W2 . relu(W1 . avgpool(y))), no bias, norm layer disabled (:36).
// avgpool comes from the conv2 epilogue's per-tile column sums.  One block per image.  w1: [mid][c], w2: [mid][c] (fc2^T).
__global__ __launch_bounds__(256) void se_finalize_kernel(const float* __restrict__ stats, int tiles, int c, int mid,
                                                          int hw, const float* __restrict__ w1,
                                                          const float* __restrict__ w2, float* __restrict__ s) {
    __shared__ float pooled[512];
    __shared__ float hid[64];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int ch = tid; ch < c; ch += 256) {
        double acc = 0.0;
        for (int t = 0; t < tiles; ++t) acc += (double)stats[(((long long)img * tiles + t) * c + ch) * 2];
        pooled[ch] = (float)(acc / hw);
    }
    __syncthreads();
    for (int m = wave; m < mid; m += 4) {
        float acc = 0.f;
        for (int ch = lane; ch < c; ch += 64) acc += w1[m * c + ch] * pooled[ch];
        acc = wave_sum(acc);
        if (lane == 0) hid[m] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += 256) {   // w2 is stored transposed, [mid][c]: coalesced across threads
        float acc = 0.f;
        for (int m = 0; m < mid; ++m) acc += w2[m * c + ch] * hid[m];
        s[(long long)img * c + ch] = 1.0f / (1.0f + expf(-acc));
    }
}

// ---- out = relu(s[n][c] * y + shortcut)   (SEBasicBlock.forward, SERes18_IBN.py:123-128)
__global__ void se_combine_kernel(const float* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ s,
                                  long long total4, int hw, int c, float* __restrict__ out) {
    const int c4n = c >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c4n);
        const long long pix = i / c4n;
        const long long img = pix / hw;
        const f32x4 yy = *(const f32x4*)(y + i * 4);
        const f32x4 rr = *(const f32x4*)(sc + i * 4);
        const f32x4 ss = *(const f32x4*)(s + img * c + cc * 4);
        f32x4 o = ss * yy + rr;
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        *(f32x4*)(out + i * 4) = o;
    }
}

// ---- SE gate + combine in ONE launch (SERes18_IBN.py:32-41 + :123-128): grid (slices, images).  Every block recomputes its
// image's gate from the conv2 epilogue's partial sums exactly as se_finalize_kernel does (a few microseconds), then streams its
// slice of the image: out = relu(gate * y + shortcut).  One launch instead of two, and enough blocks for a tracking-sized batch.
// SMALL (a tracking frame: few blocks, the launch is its dependent chain): the gate's weights - w1 rows of this wave's hidden units, the
// w2 column entries of this thread's channels - are requested BEFORE the pooled sums are formed, so the three phases of the gate wait
// for one memory round trip instead of three (layer 4: 17-18 us per launch, 12 of them the gate).  Same operations in the same order.
template <bool SMALL>
__global__ __launch_bounds__(256) void se_tail_kernel(const float* __restrict__ stats, int tiles, int c, int mid, int hw,
                                                      const float* __restrict__ w1, const float* __restrict__ w2,
                                                      const float* __restrict__ y, const float* __restrict__ sc, int rows,
                                                      float* __restrict__ out, _Float16* __restrict__ packed, int* __restrict__ fault) {
    __shared__ __attribute__((aligned(16))) float pooled[512];
    __shared__ float hid[64];
    __shared__ __attribute__((aligned(16))) float gate[512];
    const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int MJ = SMALL ? 8 : 16, MW2 = 32;      // SMALL: mid <= 32 (the launcher checks): 8 hidden units per wave, 32 w2 entries per channel
    f32x4 w1v[SMALL ? 8 : 1][2];
    float w2v[SMALL ? 2 : 1][SMALL ? MW2 : 1];
    if constexpr (SMALL) {
        const int c4v = c >> 2;
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int m = wave + 4 * j, c4 = lane + 64 * k;
                if (m < mid && c4 < c4v) w1v[j][k] = *(const f32x4*)(w1 + (long long)m * c + c4 * 4);
            }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int m = 0; m < MW2; ++m) {
                const int ch = tid + 256 * q;
                if (ch < c && m < mid) w2v[q][m] = w2[m * c + ch];
            }
    }
    for (int ch = tid; ch < c; ch += 256) {
        double acc = 0.0;
        for (int t = 0; t < tiles; ++t) acc += (double)stats[(((long long)img * tiles + t) * c + ch) * 2];
        pooled[ch] = (float)(acc / hw);
    }
    __syncthreads();
    {   // hidden units: wave w owns m = w, w + 4, ...; every load of the wave is issued before the first reduction
        // (one dependent load-reduce round per unit made this phase ~10 us at c = 512: it is what a tracking frame waits for)
        const int c4v = c >> 2;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j] = 0.f;
            if (SMALL && j >= MJ) continue;
            const int m = wave + 4 * j;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c4 = lane + 64 * k;
                if (m < mid && c4 < c4v) {
                    f32x4 wv;
                    if constexpr (SMALL) wv = w1v[j < MJ ? j : 0][k];
                    else wv = *(const f32x4*)(w1 + (long long)m * c + c4 * 4);
                    const f32x4 pv = *(const f32x4*)(pooled + c4 * 4);
                    acc[j] += wv.x * pv.x + wv.y * pv.y + wv.z * pv.z + wv.w * pv.w;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (wave + 4 * j < mid) {   // wave-uniform
                const float v = wave_sum(acc[j]);
                if (lane == 0) hid[wave + 4 * j] = fmaxf(v, 0.f);
            }
        }
    }
    __syncthreads();
    if constexpr (SMALL) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ch = tid + 256 * q;
            if (ch < c) {
                float acc = 0.f;
#pragma unroll
                for (int m = 0; m < MW2; ++m)
                    if (m < mid) acc += w2v[q][m] * hid[m];
                gate[ch] = 1.0f / (1.0f + expf(-acc));
            }
        }
    } else {
        for (int ch = tid; ch < c; ch += 256) {
            float acc = 0.f;
#pragma unroll 8
            for (int m = 0; m < mid; ++m) acc += w2[m * c + ch] * hid[m];
            gate[ch] = 1.0f / (1.0f + expf(-acc));
        }
    }
    __syncthreads();
    const int c4n = c >> 2;
    const long long base = ((long long)img * hw + (long long)blockIdx.x * rows) * c;
    const int total4 = rows * c4n;
    unsigned vm = 0u;
    for (int i = tid; i < total4; i += 256) {
        const int cc = i % c4n;
        const f32x4 yy = *(const f32x4*)(y + base + (long long)i * 4);
        const f32x4 rr = *(const f32x4*)(sc + base + (long long)i * 4);
        const f32x4 ss = *(const f32x4*)&gate[cc * 4];
        f32x4 o = ss * yy + rr;
        if (packed) vm = range_acc(range_acc(range_acc(range_acc(vm, o.x), o.y), o.z), o.w);   // before the ReLU: max(NaN, 0) is 0
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        if (out) *(f32x4*)(out + base + (long long)i * 4) = o;
        if (packed) {   // precision 2: the next block's convolutions read [oh | ol'] (ol' = f16((o - oh) 2^11)): written here, not by a pass of its own
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 hi = {(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
            const h4 lo = {(_Float16)((o.x - (float)hi.x) * 2048.0f), (_Float16)((o.y - (float)hi.y) * 2048.0f),
                           (_Float16)((o.z - (float)hi.z) * 2048.0f), (_Float16)((o.w - (float)hi.w) * 2048.0f)};
            const long long pix = (long long)img * hw + (long long)blockIdx.x * rows + i / c4n;
            *(h4*)(packed + pix * 2 * c + cc * 4) = hi;
            *(h4*)(packed + pix * 2 * c + c + cc * 4) = lo;
        }
    }
    range_raise(fault, vm);
}

// ---- GeM (attention_pooling.py:58-60) + BNNeck (SERes18_IBN.py:268).  grid (c / 64, images); 256 threads = 16 channel quads x
// 16 pixel groups (one block per image with a thread per channel walked the 128 pixels serially: 150 us for a tracking frame).
__global__ __launch_bounds__(256) void gem_neck_kernel(const float* __restrict__ x, int hw, int c,
                                                       const float* __restrict__ p_ptr, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ gem_out,
                                                       float* __restrict__ emb, int* __restrict__ fault) {
    __shared__ float part[16][64 + 1];
    const int img = blockIdx.y, c0 = blockIdx.x * 64;
    const int quad = threadIdx.x & 15, pg = threadIdx.x >> 4;
    const float p = p_ptr[0];
    const bool cube = p == 3.0f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* xi = x + (long long)img * hw * c + c0 + quad * 4;
    // x^p for a trained p (GeM's p is a parameter, initialised to 3: attention_pooling.py:58-60): exp2(p log2 x) on the transcendental
    // unit.  ocml's powf is ~50 instructions per element - 246 us per 1024-crop pass against 53 us for the p = 3 form; x >= 1e-6 and
    // p in the low single digits keep both steps in their normal range, the error (~1e-6 of a term) is below the fp32 sum's own.
    auto powp = [&](float f) { return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(f)); };
    int px = pg;
    for (; px + 7 * 16 < hw; px += 8 * 16) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(xi + (long long)(px + 16 * u) * c);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = fmaxf(v[u][e], 1e-6f);
                acc[e] += cube ? f * f * f : powp(f);
            }
    }
    for (; px < hw; px += 16) {
        const f32x4 v = *(const f32x4*)(xi + (long long)px * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f = fmaxf(v[e], 1e-6f);
            acc[e] += cube ? f * f * f : powp(f);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[pg][quad * 4 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += part[g][threadIdx.x];
        const int ch = c0 + threadIdx.x;
        const float m = t / (float)hw;
        const float g = cube ? cbrtf(m) : powf(m, 1.0f / p);
        if (gem_out) gem_out[(long long)img * c + ch] = g;
        const float ev = g * scale[ch] + shift[ch];
        emb[(long long)img * c + ch] = ev;
        if (fault && !(fabsf(ev) < INFINITY)) fault[1] = 1;   // a non-finite embedding: the context reports it (reid_ctx.fault)
    }
}

// ---- |x_i|^2 per row, one wave per row
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, int m, int d, long long ld,
                                                         float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= m) return;
    const float* xr = x + (long long)row * ld;
    float acc = 0.f;
    for (int k = lane; k < d; k += 64) acc += xr[k] * xr[k];
    acc = wave_sum(acc);
    if (lane == 0) out[row] = acc;
}

inline int grid_for(long long work, int block) {
    long long g = (work + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));  // cap at 256 CUs x 8 blocks, grid-stride the rest
}

}  // namespace

int launch_nchw_to_nhwc3(reid_ctx* ctx, const float* x, int n, int h, int w, float* out) {
    const long long npix = (long long)n * h * w;
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, npix * 24.0);
    hipLaunchKernelGGL(nchw_to_nhwc3_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, ctx->stream, x, npix, h * w, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_resize_norm(reid_ctx* ctx, const uint8_t* packed, const long long* offsets, const int* hw, int n, int H, int W,
                       int pitch, float* out) {
    const long long total = (long long)n * H * W;
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, total * 15.0);
    hipLaunchKernelGGL(resize_norm_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, packed, offsets, hw, n,
                       H, W, pitch, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_maxpool3s2(reid_ctx* ctx, const float* x, int n, int h, int w, int c, float* out) {
    ARG_CHECK(c % 4 == 0);
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)n * ho * wo * (c / 4);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, ((double)n * h * w * c + (double)n * ho * wo * c) * 4.0);
    hipLaunchKernelGGL(maxpool3s2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, x, n, h, w, c, ho, wo, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_norm_finalize(reid_ctx* ctx, const float* stats, int n_img, int tiles, int c, int half, int hw,
                         const float* in_gamma, const float* in_beta, const float* bn_scale, const float* bn_shift,
                         float* a_scale, float* a_shift) {
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n_img * c * (tiles * 8.0 + 8.0));
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(n_img), dim3(c < 256 ? c : 256), 0, ctx->stream, stats, tiles, c, half, hw,
                       in_gamma, in_beta, bn_scale, bn_shift, a_scale, a_shift);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_in_apply(reid_ctx* ctx, float* x, const float* stats, int n_img, int tiles, int c, int half, int hw,
                    const float* in_gamma, const float* in_beta) {
    ARG_CHECK(half >= 4 && half % 4 == 0 && half <= 256 && hw % 128 == 0);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n_img * hw * half * 8.0);
    hipLaunchKernelGGL(in_apply_kernel, dim3(hw / 128, n_img), dim3(256), 0, ctx->stream, x, stats, tiles, c, half, hw, 128, in_gamma,
                       in_beta);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_in_apply_pack(reid_ctx* ctx, const float* x, const float* stats, int n_img, int tiles, int c, int half, int hw,
                         const float* in_gamma, const float* in_beta, _Float16* packed, bool in_half_only) {
    ARG_CHECK(half >= 4 && half % 4 == 0 && c % 4 == 0 && c <= 512 && hw % 128 == 0 && packed);
    int rows = 128;    // few images (a tracking frame): shorter slices, so that there are blocks for every CU
    while ((long long)n_img * (hw / rows) < 512 && rows > 16) rows >>= 1;
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n_img * hw * (in_half_only ? half : c) * 8.0);
    hipLaunchKernelGGL(in_apply_pack_kernel, dim3(hw / rows, n_img), dim3(256), 0, ctx->stream, x, stats, tiles, c, half, hw, rows,
                       in_gamma, in_beta, packed, in_half_only ? 1 : 0, ctx->fault);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_se_finalize(reid_ctx* ctx, const float* stats, int n_img, int tiles, int c, int mid, int hw, const float* w1,
                       const float* w2, float* s) {
    ARG_CHECK(c <= 512 && mid <= 64);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n_img * c * (tiles * 8.0 + 4.0));
    hipLaunchKernelGGL(se_finalize_kernel, dim3(n_img), dim3(256), 0, ctx->stream, stats, tiles, c, mid, hw, w1, w2, s);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_se_combine(reid_ctx* ctx, const float* y, const float* sc, const float* s, int n_img, int hw, int c, float* out) {
    ARG_CHECK(c % 4 == 0);
    const long long total4 = (long long)n_img * hw * (c / 4);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, total4 * 48.0);
    hipLaunchKernelGGL(se_combine_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, ctx->stream, y, sc, s, total4, hw, c, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

// slices per image of the fused SE tail / in-place norm kernels: enough blocks to fill the chip, at least 16 rows per block
int tail_slices(int n_img, int hw) {
    int s = 1;
    while ((long long)n_img * s < 1024 && hw / (s * 2) >= 16 && hw % (s * 2) == 0) s *= 2;
    return s;
}

int launch_se_tail(reid_ctx* ctx, const float* stats, int n_img, int tiles, int c, int mid, int hw, const float* w1, const float* w2,
                   const float* y, const float* sc, float* out, _Float16* packed) {
    ARG_CHECK(c % 4 == 0 && c <= 512 && mid <= 64 && (out || packed));
    const int slices = tail_slices(n_img, hw);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n_img * hw * c * 12.0);
    // a tracking frame, layer 4 (64 KB of w1 + 64 KB of w2 per gate): the gate's weights requested up front, 18 -> 14 us per launch;
    // the smaller layers' launches have more blocks than gate work and lose with it (layer 1: 16 -> 31 us)
    if ((long long)slices * n_img <= 512 && c >= 512 && mid <= 32)
        hipLaunchKernelGGL(se_tail_kernel<true>, dim3(slices, n_img), dim3(256), 0, ctx->stream, stats, tiles, c, mid, hw, w1, w2, y, sc,
                           hw / slices, out, packed, ctx->fault);
    else
        hipLaunchKernelGGL(se_tail_kernel<false>, dim3(slices, n_img), dim3(256), 0, ctx->stream, stats, tiles, c, mid, hw, w1, w2, y, sc,
                           hw / slices, out, packed, ctx->fault);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_gem_neck(reid_ctx* ctx, const float* x, int n_img, int hw, int c, const float* p, const float* scale,
                    const float* shift, float* gem_out, float* emb) {
    ARG_CHECK(c % 64 == 0);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n_img * hw * c * 4.0);
    hipLaunchKernelGGL(gem_neck_kernel, dim3(c / 64, n_img), dim3(256), 0, ctx->stream, x, hw, c, p, scale, shift, gem_out, emb,
                       ctx->fault);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_row_sqnorm(reid_ctx* ctx, const float* x, int m, int d, long long ld, float* out) {
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)m * d * 4.0);
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((m + 3) / 4), dim3(256), 0, ctx->stream, x, m, d, ld, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// x[img][p][c] += coeff * table[idx[img]][c]: the optional side-information terms - SERse18_IBN's camera bias on the BNNeck
// output (SERes18_IBN.py:269-270; hw = 1, C = 512) and Swin's view embedding on the SFE output (swin_transformer.py:301-302;
// hw = H/4 * W/4, C = 96)
namespace {
__global__ void add_indexed_rows_kernel(float* __restrict__ x, long long per_img, int C, const float* __restrict__ table,
                                        const int32_t* __restrict__ idx, float coeff) {
    const int img = blockIdx.y;
    const float* row = table + (long long)idx[img] * C;
    float* xi = x + (long long)img * per_img;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_img; i += (long long)gridDim.x * blockDim.x)
        xi[i] += coeff * row[i % C];
}
}  // namespace

int launch_add_indexed_rows(reid_ctx* ctx, float* x, int n, long long hw, int C, const float* table, const int32_t* d_idx, float coeff) {
    ARG_CHECK(x && table && d_idx && n >= 1 && hw >= 1 && C >= 1);
    const long long per_img = hw * C;
    const int gx = (int)((per_img + 256 * 8 - 1) / (256 * 8));
    prof_begin(ctx, REID_K_ELEMENTWISE, (double)n * per_img * 2.0, (double)n * per_img * 8.0);
    hipLaunchKernelGGL(add_indexed_rows_kernel, dim3(gx < 1 ? 1 : gx, n), dim3(256), 0, ctx->stream, x, per_img, C, table, d_idx, coeff);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
