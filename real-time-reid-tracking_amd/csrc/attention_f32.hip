// Attention tails of the sibling backbones of SERse18_IBN, exact fp32, NHWC activations [img][h][w][c]:
//   * CARes18_IBN (reid/backbones/CARes18.py:102-162): out = relu(TripletAttention(y) + shortcut)
//     TripletAttention (triplet_attention.py:46-101): three AttentionGates - ZPool (unbiased std, mean) over C / H / W, a 7x7
//     conv (2 -> 1) over the remaining plane, BN(1), sigmoid - and out = 1/3 * (y*s_hw[h,w] + y*s_cw[c,w] + y*s_hc[h,c]).
//   * EMARes18_IBN (reid/backbones/EMA_Res18.py:10-86): out = relu(EMA(y) + shortcut), EMA with 32 channel groups.
// Everything here is reductions and elementwise work over an image that sits in L2 / Infinity Cache: HBM-bound, no MFMA.
#include "reid_internal.h"
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- TripletAttention, step 1: ZPool maps of one image.  maps = [hw: 2][H][W] | [cw: 2][C][W] | [hc: 2][H][C], channel 0 = std
// (unbiased, torch.std), channel 1 = mean.  grid (images, 3 parts); sums in fp64.
__global__ __launch_bounds__(256) void ta_stats_kernel(const float* __restrict__ y, int H, int W, int C, float* __restrict__ maps) {
    const int img = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    const int hw = H * W;
    const float* yi = y + (long long)img * hw * C;
    float* m = maps + (long long)img * 2 * (hw + C * W + H * C);
    auto put = [&](float* dst, int plane, int idx, double s1, double s2, int n) {
        const double mean = s1 / n;
        double var = (s2 - n * mean * mean) / (n - 1);
        if (var < 0.0) var = 0.0;
        dst[idx] = (float)sqrt(var);
        dst[plane + idx] = (float)mean;
    };
    if (part == 0) {          // over C for every pixel: one wave per pixel
        const int wave = tid >> 6, lane = tid & 63;
        for (int p = wave; p < hw; p += 4) {
            double s1 = 0.0, s2 = 0.0;
            for (int c = lane; c < C; c += 64) {
                const double v = yi[(long long)p * C + c];
                s1 += v;
                s2 += v * v;
            }
            s1 = wave_sum_d(s1);
            s2 = wave_sum_d(s2);
            if (lane == 0) put(m, hw, p, s1, s2, C);
        }
    } else if (part == 1) {   // over H for every (c, w)
        float* dst = m + 2 * hw;
        for (int i = tid; i < C * W; i += 256) {
            const int w = i / C, c = i - w * C;         // consecutive threads -> consecutive channels (coalesced)
            double s1 = 0.0, s2 = 0.0;
            for (int h = 0; h < H; ++h) {
                const double v = yi[((long long)h * W + w) * C + c];
                s1 += v;
                s2 += v * v;
            }
            put(dst, C * W, c * W + w, s1, s2, H);
        }
    } else {                  // over W for every (h, c)
        float* dst = m + 2 * hw + 2 * C * W;
        for (int i = tid; i < H * C; i += 256) {
            const int h = i / C, c = i - h * C;
            double s1 = 0.0, s2 = 0.0;
            for (int w = 0; w < W; ++w) {
                const double v = yi[((long long)h * W + w) * C + c];
                s1 += v;
                s2 += v * v;
            }
            put(dst, H * C, h * C + c, s1, s2, W);
        }
    }
}

// ---- step 2: gate = sigmoid(BN(conv7x7(maps))) on the three planes.  wts: [3 gates: cw, hc, hw][100] = conv weight [2][7][7],
// BN scale, BN shift.  gates = [hw][H][W] | [cw][C][W] | [hc][H][C].  grid (images, 3 parts).
__global__ __launch_bounds__(256) void ta_gate_kernel(const float* __restrict__ maps, int H, int W, int C, const float* __restrict__ wts,
                                                      float* __restrict__ gates) {
    __shared__ float wk[100];
    const int img = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    const int hw = H * W;
    int P, Q, moff, goff, gsel;
    if (part == 0) { P = H; Q = W; moff = 0; goff = 0; gsel = 2; }                                   // hw gate: plane (H, W)
    else if (part == 1) { P = C; Q = W; moff = 2 * hw; goff = hw; gsel = 0; }                        // cw gate: plane (C, W)
    else { P = H; Q = C; moff = 2 * hw + 2 * C * W; goff = hw + C * W; gsel = 1; }                   // hc gate: plane (H, C)
    if (tid < 100) wk[tid] = wts[gsel * 100 + tid];
    __syncthreads();
    const float* m = maps + (long long)img * 2 * (hw + C * W + H * C) + moff;
    float* g = gates + (long long)img * (hw + C * W + H * C) + goff;
    const int plane = P * Q;
    for (int i = tid; i < plane; i += 256) {
        const int pr = i / Q, qc = i - pr * Q;
        float acc = 0.f;
        for (int ch = 0; ch < 2; ++ch)
            for (int r = 0; r < 7; ++r) {
                const int pp = pr + r - 3;
                if ((unsigned)pp >= (unsigned)P) continue;
                for (int s = 0; s < 7; ++s) {
                    const int qq = qc + s - 3;
                    if ((unsigned)qq >= (unsigned)Q) continue;
                    acc += wk[ch * 49 + r * 7 + s] * m[ch * plane + pp * Q + qq];
                }
            }
        acc = acc * wk[98] + wk[99];
        g[i] = 1.0f / (1.0f + expf(-acc));
    }
}

// ---- step 3: out = relu(1/3 * (y*s_hw + y*s_cw + y*s_hc) + shortcut)   (triplet_attention.py:97-98, CARes18.py:155-157)
__global__ void ta_apply_kernel(const float* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ gates, long long total,
                                int H, int W, int C, float* __restrict__ out) {
    const int hw = H * W;
    const int gsz = hw + C * W + H * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long pix = i / C;
        const int p = (int)(pix % hw);
        const long long img = pix / hw;
        const int h = p / W, w = p - h * W;
        const float* g = gates + img * gsz;
        const float v = y[i];
        const float a = v * g[p], b = v * g[hw + c * W + w], d = v * g[hw + C * W + h * C + c];
        const float t = 0.3333333333333333f * ((a + b) + d);
        out[i] = fmaxf(t + sc[i], 0.f);
    }
}

// ---- EMA (EMA_Res18.py:23-38) + residual + ReLU: one block per (image, channel group), the group's cg x H x W slab in LDS.
// prm = conv1x1 w [cg][cg], b [cg] | conv3x3 w [cg][cg][3][3], b [cg] | GroupNorm weight [cg], bias [cg]
__global__ __launch_bounds__(256) void ema_tail_kernel(const float* __restrict__ y, const float* __restrict__ sc, int H, int W, int C,
                                                       const float* __restrict__ prm, float* __restrict__ out) {
    extern __shared__ float sm[];
    const int cg = C / 32, hw = H * W, HW2 = H + W;
    float* gx = sm;                    // [cg][hw]
    float* x1 = gx + cg * hw;          // [cg][hw]
    float* x2 = x1 + cg * hw;          // [cg][hw]
    float* cat = x2 + cg * hw;         // [cg][H + W]  row / column means, then their sigmoid after the 1x1 conv
    float* sig = cat + cg * HW2;       // [cg][H + W]
    float* small = sig + cg * HW2;     // mu[cg], rstd[cg], a1[cg], a2[cg], s1[cg], s2[cg]
    const int img = blockIdx.x >> 5, grp = blockIdx.x & 31, c0 = grp * cg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* w1 = prm;
    const float* b1 = w1 + cg * cg;
    const float* w3 = b1 + cg;
    const float* b3 = w3 + cg * cg * 9;
    const float* gw = b3 + cg;
    const float* gb = gw + cg;
    const long long base = (long long)img * hw * C + c0;
    for (int i = tid; i < cg * hw; i += 256) {
        const int p = i / cg, c = i - p * cg;
        gx[c * hw + p] = y[base + (long long)p * C + c];
    }
    __syncthreads();
    for (int i = tid; i < cg * HW2; i += 256) {           // pool_h (mean over w) | pool_w (mean over h)
        const int c = i / HW2, j = i - c * HW2;
        float s = 0.f;
        if (j < H) { for (int w = 0; w < W; ++w) s += gx[c * hw + j * W + w]; s /= (float)W; }
        else { for (int h = 0; h < H; ++h) s += gx[c * hw + h * W + (j - H)]; s /= (float)H; }
        cat[i] = s;
    }
    __syncthreads();
    for (int i = tid; i < cg * HW2; i += 256) {           // conv1x1 over the channels of the group, then sigmoid
        const int co = i / HW2, j = i - co * HW2;
        float acc = b1[co];
        for (int ci = 0; ci < cg; ++ci) acc += w1[co * cg + ci] * cat[ci * HW2 + j];
        sig[i] = 1.0f / (1.0f + expf(-acc));
    }
    __syncthreads();
    for (int i = tid; i < cg * hw; i += 256) {            // gated slab (before GroupNorm) and the 3x3 conv of the raw slab
        const int c = i / hw, p = i - c * hw;
        const int h = p / W, w = p - h * W;
        x1[i] = gx[i] * sig[c * HW2 + h] * sig[c * HW2 + H + w];
        float acc = b3[c];
        for (int ci = 0; ci < cg; ++ci)
            for (int r = 0; r < 3; ++r) {
                const int hh = h + r - 1;
                if ((unsigned)hh >= (unsigned)H) continue;
                for (int s = 0; s < 3; ++s) {
                    const int ww = w + s - 1;
                    if ((unsigned)ww >= (unsigned)W) continue;
                    acc += w3[((c * cg + ci) * 3 + r) * 3 + s] * gx[ci * hw + hh * W + ww];
                }
            }
        x2[i] = acc;
    }
    __syncthreads();
    for (int c = wave; c < cg; c += 4) {                  // GroupNorm statistics (one group per channel, biased variance, eps 1e-5)
        float s = 0.f;
        for (int p = lane; p < hw; p += 64) s += x1[c * hw + p];
        const float mu = wave_sum_f(s) / (float)hw;
        float q = 0.f;
        for (int p = lane; p < hw; p += 64) { const float d = x1[c * hw + p] - mu; q += d * d; }
        const float var = wave_sum_f(q) / (float)hw;
        float s2 = 0.f;
        for (int p = lane; p < hw; p += 64) s2 += x2[c * hw + p];
        s2 = wave_sum_f(s2) / (float)hw;
        if (lane == 0) {
            small[c] = mu;
            small[cg + c] = 1.0f / sqrtf(var + 1e-5f);
            small[3 * cg + c] = s2;                       // agp(x2)
        }
    }
    __syncthreads();
    for (int i = tid; i < cg * hw; i += 256) {
        const int c = i / hw;
        x1[i] = (x1[i] - small[c]) * small[cg + c] * gw[c] + gb[c];
    }
    __syncthreads();
    for (int c = wave; c < cg; c += 4) {                  // agp(x1)
        float s = 0.f;
        for (int p = lane; p < hw; p += 64) s += x1[c * hw + p];
        s = wave_sum_f(s) / (float)hw;
        if (lane == 0) small[2 * cg + c] = s;
    }
    __syncthreads();
    if (tid < 2) {                                         // softmax over the group's channels of agp(x1) / agp(x2)
        const float* a = small + (2 + tid) * cg;
        float* o = small + (4 + tid) * cg;
        float mx = -INFINITY;
        for (int c = 0; c < cg; ++c) mx = fmaxf(mx, a[c]);
        float den = 0.f;
        for (int c = 0; c < cg; ++c) { o[c] = expf(a[c] - mx); den += o[c]; }
        for (int c = 0; c < cg; ++c) o[c] /= den;
    }
    __syncthreads();
    for (int p = tid; p < hw; p += 256) {                  // weights = x11 . x2 + x21 . x1 -> out = relu(gx * sigmoid(weights) + shortcut)
        float wa = 0.f, wb = 0.f;
        for (int c = 0; c < cg; ++c) {
            wa += small[4 * cg + c] * x2[c * hw + p];
            wb += small[5 * cg + c] * x1[c * hw + p];
        }
        const float g = 1.0f / (1.0f + expf(-(wa + wb)));
        const long long o = base + (long long)p * C;
        for (int c = 0; c < cg; ++c) out[o + c] = fmaxf(gx[c * hw + p] * g + sc[o + c], 0.f);
    }
}

inline int grid_for(long long work, int block) {
    long long g = (work + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

int launch_ta_tail(reid_ctx* ctx, const float* y, const float* sc, int n_img, int H, int W, int C, const float* wts, float* out) {
    const long long per = (long long)H * W + (long long)C * W + (long long)H * C;
    float *maps, *gates;
    REID_TRY(ctx_ws(ctx, "ta.maps", (size_t)n_img * 2 * per * 4, (void**)&maps));
    REID_TRY(ctx_ws(ctx, "ta.gates", (size_t)n_img * per * 4, (void**)&gates));
    const long long total = (long long)n_img * H * W * C;
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)total * 4.0 * 6.0);
    hipLaunchKernelGGL(ta_stats_kernel, dim3(n_img, 3), dim3(256), 0, ctx->stream, y, H, W, C, maps);
    hipLaunchKernelGGL(ta_gate_kernel, dim3(n_img, 3), dim3(256), 0, ctx->stream, maps, H, W, C, wts, gates);
    hipLaunchKernelGGL(ta_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, y, sc, gates, total, H, W, C, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

int launch_ema_tail(reid_ctx* ctx, const float* y, const float* sc, int n_img, int H, int W, int C, const float* prm, float* out) {
    ARG_CHECK(C % 32 == 0 && C / 32 <= 16);
    const int cg = C / 32, hw = H * W;
    const size_t lds = ((size_t)3 * cg * hw + (size_t)2 * cg * (H + W) + 6 * cg) * 4;
    ARG_CHECK(lds <= 150 * 1024);
    // per device and not worth caching: a process-wide flag left a second device (or a racing thread) with the 48 KB default
    HIP_TRY(hipFuncSetAttribute((const void*)ema_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    prof_begin(ctx, REID_K_ELEMENTWISE, 0, (double)n_img * hw * C * 12.0);
    hipLaunchKernelGGL(ema_tail_kernel, dim3(n_img * 32), dim3(256), lds, ctx->stream, y, sc, H, W, C, prm, out);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
