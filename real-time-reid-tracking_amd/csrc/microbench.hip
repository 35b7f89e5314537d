// Operand-feed microbenchmarks (experiments, not part of the public header): how many bytes per second can a CU pull
// into LDS (global_load_lds) or into registers (global_load_dwordx4) from a buffer of a given footprint, 8 waves/block?
#include "reid_internal.h"

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

}  // namespace

extern "C" int reid_debug_feed(reid_ctx* ctx, int mode, size_t footprint, int rowb, size_t stride, int iters, int inflight,
                               float* gbs_per_cu, float* tbs_chip) {
    ARG_CHECK(ctx && gbs_per_cu && tbs_chip && (rowb == 64 || rowb == 128 || rowb == 256 || rowb == 1024));
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
