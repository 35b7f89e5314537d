// Operand-feed microbenchmarks (experiments, not part of the public header): how many bytes per second can a CU pull
// into LDS (global_load_lds) or into registers (global_load_dwordx4) from a buffer of a given footprint, 8 waves/block?
#include "reid_internal.h"
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(uintptr_t)(p))

namespace {

// mode 0: LDS-DMA, 1 KiB per wave-instruction, rows of `rowb` bytes gathered from `stride`-spaced rows
// mode 1: same addresses into registers (global_load_dwordx4), accumulated so the loads stay live
template <int MODE>
__global__ __launch_bounds__(512) void feed_kernel(const char* __restrict__ src, size_t footprint, int iters, int rowb,
                                                   size_t stride, int inflight, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) char lds[128 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ch = rowb / 16;                      // chunks per row
    const int rows_per_inst = 64 / ch;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    size_t base = ((size_t)blockIdx.x * 8 + wave) * 1024 * 131;   // de-correlate blocks
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int j = 0; j < 8; ++j) {
            const size_t row = (base / rowb + (size_t)(it * 8 + j) * rows_per_inst + lane / ch);
            const size_t off = (row * stride + (lane % ch) * 16) % footprint;
            if constexpr (MODE == 0) {
                __builtin_amdgcn_global_load_lds(GPTR(src + off), LPTR(lds + (wave * 16 + (j + 8 * (it & 1))) * 1024), 16, 0, 0);
            } else {
                const f32x4 v = *(const f32x4*)(src + off);
                acc += v;
            }
        }
        if constexpr (MODE == 0) {
            if (inflight == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if constexpr (MODE == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc.x = ((const float*)lds)[threadIdx.x];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

// MFMA-shape experiment (MI355X_MICROARCH.md, DVFS give-back item 7): the inner loop of the halo convolution - per wave and
// per 32 k: A and B fragments re-read from LDS with ds_read_b128, then the MFMAs of a 64 x 64 wave tile - built once with
// v_mfma_f32_32x32x16_f16 (2 x (2A + 2B reads, 4 MFMAs)) and once with v_mfma_f32_16x16x32_f16 (4A + 4B reads, 16 MFMAs):
// same FLOPs, same LDS bytes, same accumulator registers.  LDS holds pseudo-random f16 values (the chip is clock-limited under
// MFMA load on random data); results are summed into a sink so that nothing is optimised away.
typedef _Float16 f16mb;
typedef f16mb half8mb __attribute__((ext_vector_type(8)));
typedef float f32x16mb __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_shape_kernel(int iters, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned r = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = tid; i < 64 * 1024 / 2; i += 512) {
        r = r * 1664525u + 1013904223u;
        ((f16mb*)lds)[i] = (f16mb)(((int)(r >> 9) & 0xffff) / 32768.0f - 1.0f);
    }
    __syncthreads();
    const char* base = lds + (wave & 3) * 8192;
    float total = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16mb acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int off = ((it * 2 + kk) & 7) * 1024 + (lane >> 5) * 16;
                half8mb af[2], bf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) af[a] = *(const half8mb*)(base + (a * 32 + (lane & 31)) * 32 % 8192 + off % 4096);
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b] = *(const half8mb*)(base + 32768 + (b * 32 + (lane & 31)) * 32 % 8192 + off % 4096);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) total += acc[a][b][e];
    } else {
        typedef float f32x4mb __attribute__((ext_vector_type(4)));
        f32x4mb acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f32x4mb{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            const int off = (it & 7) * 1024 + (lane >> 4) * 16;
            half8mb af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = *(const half8mb*)(base + (a * 16 + (lane & 15)) * 64 % 8192 + off % 4096);
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = *(const half8mb*)(base + 32768 + (b * 16 + (lane & 15)) * 64 % 8192 + off % 4096);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) total += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    }
    if (total == 12345.678f) sink[0] = total;
}

}  // namespace

// Registers-only MFMA loop: what the matrix pipe sustains on THIS device with nothing else going on - no LDS, no memory.  On random
// operands the chip does not hold its clock under f16 MFMAs (round 5, tools/probes/overlap.hip: 1.55-1.6 PF for 32x32x16, 1.95-2.0 PF
// for 16x16x32 against 2.3 PF on all-zero operands and 2.5 PF nominal), and devices of the pool differ by ~10 %: bench.py reports this
// number beside the nominal peak.  512 blocks x 4 waves (two waves per SIMD), 16 independent accumulator tiles per wave.
namespace {
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void mfma_bare_kernel(int iters, int zero, float* __restrict__ sink) {
    const int tid = threadIdx.x;
    unsigned r = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    half8mb fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            r = r * 1664525u + 1013904223u;
            fa[i][j] = zero ? (f16mb)0.f : (f16mb)(((int)(r >> 9) & 0xffff) / 32768.0f - 1.0f);
            r = r * 1664525u + 1013904223u;
            fb[i][j] = zero ? (f16mb)0.f : (f16mb)(((int)(r >> 9) & 0xffff) / 32768.0f - 1.0f);
        }
    float total = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16mb acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a & 1], fb[a >> 1], acc[a], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) total += acc[a][e];
    } else {
        typedef float f32x4mb2 __attribute__((ext_vector_type(4)));
        f32x4mb2 acc[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) acc[a] = f32x4mb2{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int a = 0; a < 16; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a & 1], fb[(a >> 1) & 1], acc[a], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 16; ++a) total += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    }
    if (total == 123.456f) sink[tid] = total;
}
}  // namespace

// shape 32 (v_mfma_f32_32x32x16_f16) or 16 (v_mfma_f32_16x16x32_f16); zero != 0: all-zero operands; *tflops over ~`iters` x 16 (32) MFMAs per wave
extern "C" int reid_debug_mfma_bare(reid_ctx* ctx, int shape, int zero, int iters, float* tflops) {
    ARG_CHECK(ctx && tflops && (shape == 32 || shape == 16) && iters > 0);
    CTX_GUARD(ctx);
    float* sink;
    REID_TRY(ctx_ws(ctx, "dbg.sink", 4096, (void**)&sink));
    const int blocks = 512;
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) REID_TRY(reid_timer_start(ctx));
        if (shape == 32) hipLaunchKernelGGL(mfma_bare_kernel<32>, dim3(blocks), dim3(256), 0, ctx->stream, iters, zero, sink);
        else hipLaunchKernelGGL(mfma_bare_kernel<16>, dim3(blocks), dim3(256), 0, ctx->stream, iters, zero, sink);
    }
    float ms = 0.f;
    REID_TRY(reid_timer_stop(ctx, &ms));
    LAUNCH_CHECK();
    // per wave and iteration: 16 x 32x32x16 or 32 x 16x16x32 = 16 x 32768 multiply-adds
    *tflops = (float)((double)blocks * 4 * iters * 16.0 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12);
    return REID_OK;
}

// shape 32 or 16; returns TFLOP/s of the chip over a launch of `iters` 32-k steps per wave (64 x 64 wave tile, 8 waves, 256+ blocks)
extern "C" int reid_debug_mfma_shape(reid_ctx* ctx, int shape, int iters, int blocks, float* tflops) {
    ARG_CHECK(ctx && tflops && (shape == 32 || shape == 16) && iters > 0 && blocks > 0);
    CTX_GUARD(ctx);
    float* sink;
    REID_TRY(ctx_ws(ctx, "dbg.sink", 64, (void**)&sink));
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) REID_TRY(reid_timer_start(ctx));
        if (shape == 32) hipLaunchKernelGGL(mfma_shape_kernel<32>, dim3(blocks), dim3(512), 0, ctx->stream, iters, sink);
        else hipLaunchKernelGGL(mfma_shape_kernel<16>, dim3(blocks), dim3(512), 0, ctx->stream, iters, sink);
    }
    float ms = 0.f;
    REID_TRY(reid_timer_stop(ctx, &ms));
    LAUNCH_CHECK();
    *tflops = (float)((double)blocks * 8 * iters * 2.0 * 64 * 64 * 32 / (ms * 1e-3) / 1e12);
    return REID_OK;
}

extern "C" int reid_debug_feed(reid_ctx* ctx, int mode, size_t footprint, int rowb, size_t stride, int iters, int inflight,
                               float* gbs_per_cu, float* tbs_chip) {
    ARG_CHECK(ctx && gbs_per_cu && tbs_chip && (rowb == 64 || rowb == 128 || rowb == 256 || rowb == 1024));
    CTX_GUARD(ctx);
    char* buf;
    float* sink;
    REID_TRY(ctx_ws(ctx, "dbg.feed", footprint + 4096, (void**)&buf));
    REID_TRY(ctx_ws(ctx, "dbg.sink", 64, (void**)&sink));
    const int blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) REID_TRY(reid_timer_start(ctx));
        if (mode == 0) hipLaunchKernelGGL(feed_kernel<0>, dim3(blocks), dim3(512), 0, ctx->stream, buf, footprint, iters, rowb, stride, inflight, sink);
        else hipLaunchKernelGGL(feed_kernel<1>, dim3(blocks), dim3(512), 0, ctx->stream, buf, footprint, iters, rowb, stride, inflight, sink);
    }
    float ms = 0.f;
    REID_TRY(reid_timer_stop(ctx, &ms));
    LAUNCH_CHECK();
    const double bytes = (double)blocks * 8 * iters * 8 * 1024.0;
    *tbs_chip = (float)(bytes / (ms * 1e-3) / 1e12);
    *gbs_per_cu = (float)(bytes / (ms * 1e-3) / 1e9 / 256.0);
    return REID_OK;
}

// Co-issue experiment for the exact-fp32 path: what does a wave that stages data (VALU / LDS writes / global loads) get to
// issue while the OTHER wave of its SIMD runs back-to-back v_mfma_f32_32x32x2_f32?  512-thread blocks, one per CU: waves 0-3
// run `iters` x 16 MFMAs (4 accumulators), waves 4-7 run `iters` x 64 operations of `mode` (0 v_fma_f32, 1 ds_write_b128,
// 2 global_load_dwordx4 + wait every 8, 3 v_fma with s_setprio 3).  `roles`: bit 0 = MFMA waves active, bit 1 = other waves active,
// bit 2 = the MFMA waves issue v_mfma_f32_32x32x16_f16 instead.
// Output: median cycles per wave of each role.
namespace {
typedef float f32x16cb __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512) void coissue_kernel(int iters, int roles, const float* __restrict__ src,
                                                      unsigned long long* __restrict__ out, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float lds[512 * 4 * 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mf = wave < 4;
    float keep = 0.f;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (mf) {
        if (roles & 1) {
            f32x16cb acc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
            const float av = src[lane], bv = src[64 + lane];
            if (roles & 4) {   // the same experiment beside v_mfma_f32_32x32x16_f16 (8 passes instead of 16)
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                h8 ah, bh;
#pragma unroll
                for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)av; bh[e] = (_Float16)bv; }
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j & 3], 0, 0, 0);
                }
            } else {
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j & 3], 0, 0, 0);
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) keep += acc[a][0] + acc[a][7];
        }
    } else if (roles & 2) {
        if constexpr (MODE == 3) __builtin_amdgcn_s_setprio(3);
        if constexpr (MODE == 0 || MODE == 3) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = src[lane + j];
            const float a = src[200 + lane], b = src[300 + lane];
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 64; ++j) x[j & 7] = __builtin_fmaf(x[j & 7], a, b);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) keep += x[j];
        } else if constexpr (MODE == 1) {
            f32x4 v = {src[lane], src[lane + 1], src[lane + 2], src[lane + 3]};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    *(f32x4*)&lds[(tid * 2 + (j & 1)) * 4] = v;
                    asm volatile("" ::: "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            keep += lds[tid * 8];
        } else {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    const f32x4 v = *(const f32x4*)(src + ((size_t)(blockIdx.x * 512 + tid) * 4 + (size_t)((it * 64 + j) & 1023) * 65536));
                    s += v;
                }
            }
            keep += s.x + s.y + s.z + s.w;
        }
        if constexpr (MODE == 3) __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (keep == 12345.678f) sink[0] = keep;
}
}  // namespace

extern "C" int reid_debug_coissue(reid_ctx* ctx, int mode, int iters, int roles, double* cyc_mfma_wave, double* cyc_other_wave) {
    ARG_CHECK(ctx && mode >= 0 && mode <= 3 && iters > 0 && cyc_mfma_wave && cyc_other_wave);
    CTX_GUARD(ctx);
    float* src;
    unsigned long long* out;
    float* sink;
    const int blocks = 256;
    const size_t src_bytes = (size_t)1024 * 65536 * 4 + (size_t)blocks * 512 * 16 + 4096;
    REID_TRY(ctx_ws(ctx, "dbg.co_src", src_bytes, (void**)&src));
    REID_TRY(ctx_ws(ctx, "dbg.co_out", blocks * 8 * 8, (void**)&out));
    REID_TRY(ctx_ws(ctx, "dbg.sink", 64, (void**)&sink));
    HIP_TRY(hipMemsetAsync(src, 0x3c, src_bytes, ctx->stream));   // 0x3c3c3c3c = 0.0115 as fp32
    for (int rep = 0; rep < 2; ++rep) {
        switch (mode) {
            case 0: hipLaunchKernelGGL(coissue_kernel<0>, dim3(blocks), dim3(512), 0, ctx->stream, iters, roles, src, out, sink); break;
            case 1: hipLaunchKernelGGL(coissue_kernel<1>, dim3(blocks), dim3(512), 0, ctx->stream, iters, roles, src, out, sink); break;
            case 2: hipLaunchKernelGGL(coissue_kernel<2>, dim3(blocks), dim3(512), 0, ctx->stream, iters, roles, src, out, sink); break;
            default: hipLaunchKernelGGL(coissue_kernel<3>, dim3(blocks), dim3(512), 0, ctx->stream, iters, roles, src, out, sink); break;
        }
    }
    LAUNCH_CHECK();
    std::vector<unsigned long long> h(blocks * 8);
    HIP_TRY(hipMemcpyAsync(h.data(), out, blocks * 8 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<double> a, b;
    for (int i = 0; i < blocks; ++i)
        for (int w = 0; w < 8; ++w) (w < 4 ? a : b).push_back((double)h[i * 8 + w]);
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    *cyc_mfma_wave = a[a.size() / 2];
    *cyc_other_wave = b[b.size() / 2];
    return REID_OK;
}
