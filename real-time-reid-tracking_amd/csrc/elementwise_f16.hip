// fp16-storage variants of the HBM / Infinity-Cache-bound kernels of the embed path (fp32 math, 16-byte accesses).
#include "reid_internal.h"
#include <math.h>

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef f16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

inline int grid_for(long long work, int block) {
    long long g = (work + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// uint8 NHWC crops -> (v/255-0.5)/0.5 (feature_extractor.py:41-46) as f16 in a zero-padded NHWC4 image
// [n][hp][wp][4] (3 rows/cols of zeros before, the rest after; channel 3 = 0) that the stem GEMM gathers from
// with aligned 16-byte loads and no bounds checks.
template <typename SRC>
__global__ void prep_pad_f16_kernel(const SRC* __restrict__ x, int n, int h, int w, int hp, int wp, f16* __restrict__ out) {
    const long long total = (long long)n * hp * wp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int px = (int)(i % wp);
        const long long t = i / wp;
        const int py = (int)(t % hp);
        const int img = (int)(t / hp);
        const int iy = py - 3, ix = px - 3;
        half4 v = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) {
            const SRC* s = x + (((long long)img * h + iy) * w + ix) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float f;
                if constexpr (sizeof(SRC) == 1) f = ((float)s[c] / 255.0f - 0.5f) / 0.5f;
                else f = (float)s[c];
                v[c] = (f16)f;
            }
        }
        *(half4*)(out + i * 4) = v;
    }
}

__global__ void maxpool3s2_f16_kernel(const f16* __restrict__ x, int n, int h, int w, int c, int ho, int wo,
                                      f16* __restrict__ out) {
    const int c8n = c >> 3;
    const long long total = (long long)n * ho * wo * c8n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8n);
        long long t = i / c8n;
        const int ox = (int)(t % wo);
        t /= wo;
        const int oy = (int)(t % ho);
        const int img = (int)(t / ho);
        half8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (f16)(-65504.f);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)w) continue;
                const half8 v = *(const half8*)(x + (((long long)img * h + iy) * w + ix) * c + cc * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        }
        *(half8*)(out + (((long long)img * ho + oy) * wo + ox) * c + cc * 8) = m;
    }
}

// x <- relu(a[img][c] * x + b[img][c]) in place: IBN / BN + ReLU between conv1 and conv2 (SERes18_IBN.py:88-93)
__global__ void affine_relu_f16_kernel(f16* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                       long long total8, int hw, int c) {
    const int c8n = c >> 3;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8n);
        const long long img = (i / c8n) / hw;
        half8 v = *(const half8*)(x + i * 8);
        const float* ap = a + img * c + cc * 8;
        const float* bp = b + img * c + cc * 8;
        const f32x4 a0 = *(const f32x4*)ap, a1 = *(const f32x4*)(ap + 4);
        const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (f16)fmaxf((float)v[e] * a0[e] + b0[e], 0.f);
            v[e + 4] = (f16)fmaxf((float)v[e + 4] * a1[e] + b1[e], 0.f);
        }
        *(half8*)(x + i * 8) = v;
    }
}

__global__ void se_combine_f16_kernel(const f16* __restrict__ y, const f16* __restrict__ sc, const float* __restrict__ s,
                                      long long total8, int hw, int c, f16* __restrict__ out) {
    const int c8n = c >> 3;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8n);
        const long long img = (i / c8n) / hw;
        const half8 yy = *(const half8*)(y + i * 8);
        const half8 rr = *(const half8*)(sc + i * 8);
        const float* sp = s + img * c + cc * 8;
        const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
        half8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (f16)fmaxf(s0[e] * (float)yy[e] + (float)rr[e], 0.f);
            o[e + 4] = (f16)fmaxf(s1[e] * (float)yy[e + 4] + (float)rr[e + 4], 0.f);
        }
        *(half8*)(out + i * 8) = o;
    }
}

// GeM + BNNeck from f16 activations: grid = (c/64, n_img); 256 threads = 8 channel-octets x 32 pixel groups
__global__ __launch_bounds__(256) void gem_neck_f16_kernel(const f16* __restrict__ x, int hw, int c,
                                                           const float* __restrict__ p_ptr, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ gem_out,
                                                           float* __restrict__ emb, int* __restrict__ fault) {
    __shared__ float part[32][64 + 1];
    const int img = blockIdx.y, c0 = blockIdx.x * 64;
    const int oct = threadIdx.x & 7, pg = threadIdx.x >> 3;
    const float p = p_ptr[0];
    const bool cube = p == 3.0f;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const f16* xi = x + (long long)img * hw * c + c0 + oct * 8;
    for (int px = pg; px < hw; px += 32) {
        const half8 v = *(const half8*)(xi + (long long)px * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = fmaxf((float)v[e], 1e-6f);
            acc[e] += cube ? f * f * f : powf(f, p);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[pg][oct * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += part[g][threadIdx.x];
        const int ch = c0 + threadIdx.x;
        const float m = t / (float)hw;
        const float g = cube ? cbrtf(m) : powf(m, 1.0f / p);
        if (gem_out) gem_out[(long long)img * c + ch] = g;
        const float ev = g * scale[ch] + shift[ch];
        emb[(long long)img * c + ch] = ev;
        if (fault && !(fabsf(ev) < INFINITY)) fault[1] = 1;   // a non-finite embedding (f16 overflow upstream): reid_ctx.fault
    }
}

__global__ void f32_to_f16_kernel(const float* __restrict__ x, size_t n, f16* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (f16)x[i];
}

// stem weights [64][8][24] (k = r*24 + s*3 + c) -> [64][8][8][4] (k = r*32 + s*4 + c), zero padded
__global__ void stem_w16_kernel(const float* __restrict__ w, f16* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 256) return;
    const int co = i >> 8, k = i & 255;
    const int r = k >> 5, s = (k >> 2) & 7, c = k & 3;
    float v = 0.f;
    if (r < 7 && s < 7 && c < 3) v = w[co * 192 + r * 24 + s * 3 + c];
    out[i] = (f16)v;
}

}  // namespace

int launch_prep_u8_pad_f16(reid_ctx* ctx, const uint8_t* crops, int n, int h, int w, int hp, int wp, f16* out) {
    const long long total = (long long)n * hp * wp;
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n * h * w * 3.0 + total * 8.0);
    hipLaunchKernelGGL((prep_pad_f16_kernel<uint8_t>), dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, crops, n, h, w,
                       hp, wp, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
int launch_prep_f32_pad_f16(reid_ctx* ctx, const float* x, int n, int h, int w, int hp, int wp, f16* out) {
    const long long total = (long long)n * hp * wp;
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n * h * w * 12.0 + total * 8.0);
    hipLaunchKernelGGL((prep_pad_f16_kernel<float>), dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, x, n, h, w, hp, wp,
                       out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
int launch_maxpool3s2_f16(reid_ctx* ctx, const f16* x, int n, int h, int w, int c, f16* out) {
    ARG_CHECK(c % 8 == 0);
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)n * ho * wo * (c / 8);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, ((double)n * h * w * c + (double)n * ho * wo * c) * 2.0);
    hipLaunchKernelGGL(maxpool3s2_f16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, x, n, h, w, c, ho, wo, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
int launch_affine_relu_f16(reid_ctx* ctx, f16* x, const float* a, const float* b, int n_img, int hw, int c) {
    ARG_CHECK(c % 8 == 0);
    const long long total8 = (long long)n_img * hw * (c / 8);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, total8 * 32.0);
    hipLaunchKernelGGL(affine_relu_f16_kernel, dim3(grid_for(total8, 256)), dim3(256), 0, ctx->stream, x, a, b, total8, hw, c);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
// IBN / BN finalisation + application in ONE launch (fp16 path, SERes18_IBN.py:88-93): grid (slices, images).  Every block
// derives its image's per-channel (a, b) as norm_finalize_kernel does (InstanceNorm half from the conv epilogue's partial sums,
// BatchNorm half folded), then rewrites its rows in place: x = relu(x * a + b).
__global__ __launch_bounds__(256) void norm_apply_f16_kernel(f16* __restrict__ x, const float* __restrict__ stats, int tiles, int c,
                                                             int half, int hw, int rows, const float* __restrict__ in_gamma,
                                                             const float* __restrict__ in_beta, const float* __restrict__ bn_scale,
                                                             const float* __restrict__ bn_shift) {
    __shared__ float sa[512], sb[512];
    const int img = blockIdx.y, tid = threadIdx.x;
    for (int ch = tid; ch < c; ch += 256) {
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
        sa[ch] = a;
        sb[ch] = b;
    }
    __syncthreads();
    const int c8n = c >> 3;
    f16* base = x + ((long long)img * hw + (long long)blockIdx.x * rows) * c;
    for (int i = tid; i < rows * c8n; i += 256) {
        const int cc = i % c8n;
        half8 v = *(const half8*)(base + (long long)i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (f16)fmaxf((float)v[e] * sa[cc * 8 + e] + sb[cc * 8 + e], 0.f);
        *(half8*)(base + (long long)i * 8) = v;
    }
}

int launch_norm_apply_f16(reid_ctx* ctx, f16* x, const float* stats, int n_img, int tiles, int c, int half, int hw,
                          const float* in_gamma, const float* in_beta, const float* bn_scale, const float* bn_shift) {
    ARG_CHECK(c % 8 == 0 && c <= 512);
    const int slices = tail_slices(n_img, hw);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n_img * hw * c * 4.0);
    hipLaunchKernelGGL(norm_apply_f16_kernel, dim3(slices, n_img), dim3(256), 0, ctx->stream, x, stats, tiles, c, half, hw, hw / slices,
                       in_gamma, in_beta, bn_scale, bn_shift);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

// SE gate + combine of one SE block in one launch (SERes18_IBN.py:32-41 + :120-128): block = image.  Phase 1 is se_finalize
// (pooled mean from the conv2 epilogue's partial sums -> 8..32 hidden units -> sigmoid gate, kept in LDS), phase 2 streams the
// image: out = relu(gate * y + shortcut).  Several blocks per CU overlap one image's gate with another's streaming.
__global__ __launch_bounds__(256) void se_tail_f16_kernel(const float* __restrict__ stats, int tiles, int c, int mid, int hw,
                                                          const float* __restrict__ w1, const float* __restrict__ w2t,
                                                          const f16* __restrict__ y, const f16* __restrict__ sc,
                                                          int rows, f16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float pooled[512];
    __shared__ float hid[64];
    __shared__ float gate[512];
    const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // grid (slices, images)
    for (int ch = tid; ch < c; ch += 256) {
        double acc = 0.0;
        for (int t = 0; t < tiles; ++t) acc += (double)stats[(((long long)img * tiles + t) * c + ch) * 2];
        pooled[ch] = (float)(acc / hw);
    }
    __syncthreads();
    {   // hidden units: wave w owns m = w, w + 4, ...; every load of the wave is issued before the first reduction
        // (one dependent load-reduce round per unit made this phase 10 us at c = 512, longer than the streaming)
        const int c4n = c >> 2;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j] = 0.f;
            const int m = wave + 4 * j;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c4 = lane + 64 * k;
                if (m < mid && c4 < c4n) {
                    const float4 wv = *(const float4*)(w1 + (long long)m * c + c4 * 4);
                    const float4 pv = *(const float4*)(pooled + c4 * 4);
                    acc[j] += wv.x * pv.x + wv.y * pv.y + wv.z * pv.z + wv.w * pv.w;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (wave + 4 * j < mid) {   // wave-uniform
                float v = acc[j];
                for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
                if (lane == 0) hid[wave + 4 * j] = fmaxf(v, 0.f);
            }
        }
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += 256) {
        float acc = 0.f;
#pragma unroll 8
        for (int m = 0; m < mid; ++m) acc += w2t[m * c + ch] * hid[m];
        gate[ch] = 1.0f / (1.0f + expf(-acc));
    }
    __syncthreads();
    const int c8n = c >> 3;
    const long long base = ((long long)img * hw + (long long)blockIdx.x * rows) * c;
    const int total8 = rows * c8n;
    for (int i = tid; i < total8; i += 256) {
        const int cc = i % c8n;
        const half8 yy = *(const half8*)(y + base + (long long)i * 8);
        const half8 rr = *(const half8*)(sc + base + (long long)i * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)fmaxf(gate[cc * 8 + e] * (float)yy[e] + (float)rr[e], 0.f);
        *(half8*)(out + base + (long long)i * 8) = o;
    }
}

int launch_se_tail_f16(reid_ctx* ctx, const float* stats, int n_img, int tiles, int c, int mid, int hw, const float* w1,
                       const float* w2t, const f16* y, const f16* sc, f16* out) {
    ARG_CHECK(c % 8 == 0 && c <= 512 && mid <= 64);
    const int slices = tail_slices(n_img, hw);   // one slice per image at 1024 crops; 8-16 for a tracking frame
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n_img * hw * c * 6.0);
    hipLaunchKernelGGL(se_tail_f16_kernel, dim3(slices, n_img), dim3(256), 0, ctx->stream, stats, tiles, c, mid, hw, w1, w2t, y, sc,
                       hw / slices, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_se_combine_f16(reid_ctx* ctx, const f16* y, const f16* sc, const float* s, int n_img, int hw, int c, f16* out) {
    ARG_CHECK(c % 8 == 0);
    const long long total8 = (long long)n_img * hw * (c / 8);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, total8 * 48.0);
    hipLaunchKernelGGL(se_combine_f16_kernel, dim3(grid_for(total8, 256)), dim3(256), 0, ctx->stream, y, sc, s, total8, hw, c,
                       out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
int launch_gem_neck_f16(reid_ctx* ctx, const f16* x, int n_img, int hw, int c, const float* p, const float* scale,
                        const float* shift, float* gem_out, float* emb) {
    ARG_CHECK(c % 64 == 0);
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n_img * hw * c * 2.0);
    hipLaunchKernelGGL(gem_neck_f16_kernel, dim3(c / 64, n_img), dim3(256), 0, ctx->stream, x, hw, c, p, scale, shift, gem_out,
                       emb, ctx->fault);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
int launch_f32_to_f16(reid_ctx* ctx, const float* x, size_t n, f16* out) {
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(grid_for((long long)n, 256)), dim3(256), 0, ctx->stream, x, n, out);
    LAUNCH_CHECK();
    return REID_OK;
}
int launch_stem_w16(reid_ctx* ctx, const float* w, f16* out) {
    hipLaunchKernelGGL(stem_w16_kernel, dim3(64), dim3(256), 0, ctx->stream, w, out);
    LAUNCH_CHECK();
    return REID_OK;
}
