// Do matrix instructions, LDS fragment reads and global->LDS DMA of DIFFERENT waves overlap on a CU, on random data?
// (round 5: the phases of both halo-convolution kernels ADD - MFMA 410 + reads 249 + DMA 356 + epilogue 328 us ~ 1 416 us full.)
// 512-thread blocks, one per CU (grid 256) or two (grid 512, 256 threads); waves take roles by number:
//   M: 16 x v_mfma_f32_32x32x16_f16 per step (registers only)      R: 12 x ds_read_b128 per step      D: 3 x 1-KB global_load_lds per step
// the per-step counts are the convolution's (per wave and tile).  hipcc --offload-arch=gfx950 -O3 tools/probes/overlap.hip -o tools/probes/overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(uintptr_t)(p))

// role of wave w: bits of `roles` >> (4 * (w & 7)): 1 = M, 2 = R, 4 = D (a wave may hold several: done in sequence per step)
__global__ __launch_bounds__(512) void overlap_kernel(int steps, unsigned long long roles, const f16* __restrict__ src, size_t src_elems,
                                                      float* __restrict__ sink, int zero) {
    __shared__ __attribute__((aligned(16))) char lds[96 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = (int)((roles >> (4 * wave)) & 15);
    unsigned r = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = tid; i < 96 * 1024 / 2; i += blockDim.x) {
        r = r * 1664525u + 1013904223u;
        f16 v = (f16)(((int)(r >> 9) & 0xffff) / 32768.0f - 1.0f);
        if (zero == 1) v = (f16)0.f;
        if (zero >= 2) {   // keep only the top (zero - 2) mantissa bits: do sparse mantissas draw less power?
            unsigned short b = __builtin_bit_cast(unsigned short, v);
            b &= (unsigned short)(0xffff << (10 - (zero - 2)));
            v = __builtin_bit_cast(f16, b);
        }
        ((f16*)lds)[i] = v;
    }
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc4[16];
    for (int a = 0; a < 16; ++a) acc4[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    half8 fa[2], fb[2];
    for (int i = 0; i < 2; ++i) {
        fa[i] = *(const half8*)(lds + (lane * 16 + i * 1024));
        fb[i] = *(const half8*)(lds + 8192 + (lane * 16 + i * 1024));
    }
    half8 rd[12];
    const unsigned lbase = (unsigned)(uintptr_t)lds + 16384 + wave * 4096 + lane * 16;
    const f16* g = src + ((size_t)blockIdx.x * 8 + wave) * 8192 + lane * 8;
    float keep = 0.f;
    for (int s = 0; s < steps; ++s) {
        if (role & 4) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                __builtin_amdgcn_global_load_lds(GPTR(g + (((size_t)s * 3 + j) * 512) % 8192), LPTR(lds + 49152 + wave * 4096 + j * 1024), 16, 0, 0);
            if ((s & 1) == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        if (role & 2) {
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[j]) : "v"(lbase), "n"(j * 1024 % 4096 + (j / 4) * 16));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("" ::"v"(rd[j]));
        }
        if (role & 8) {   // the same flops as role 1 in v_mfma_f32_16x16x32_f16 (32 per step)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int a = 0; a < 16; ++a) acc4[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a & 1], fb[(a >> 1) & 1], acc4[a], 0, 0, 0);
        }
        if (role & 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a & 1], fb[a >> 1], acc[a], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) keep += acc[a][e];
    for (int a = 0; a < 16; ++a) keep += acc4[a][0] + acc4[a][1] + acc4[a][2] + acc4[a][3];
    if (keep == 123.456f) sink[tid] = keep;
}

static float run(int grid, int threads, int steps, unsigned long long roles, const f16* src, size_t n, float* sink, int zero) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(overlap_kernel, dim3(grid), dim3(threads), 0, 0, steps, roles, src, n, sink, zero);
    CK(hipEventRecord(a));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(overlap_kernel, dim3(grid), dim3(threads), 0, 0, steps, roles, src, n, sink, zero);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / 3 * 1000.f;
}

int main() {
    const size_t n = (size_t)512 * 8 * 8192;
    f16* src;
    float* sink;
    CK(hipMalloc(&src, n * 2));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(src, 0x3c, n * 2));
    const int steps = 4000;
    struct { const char* name; unsigned long long roles; } cfg[] = {
        {"M in waves 0-3 (one per SIMD)", 0x00001111ull},
        {"M in all 8 waves", 0x11111111ull},
        {"R in waves 4-7", 0x22220000ull},
        {"R in all 8 waves", 0x22222222ull},
        {"D in waves 4-7", 0x44440000ull},
        {"D in all 8 waves", 0x44444444ull},
        {"M 0-3 | R 4-7", 0x22221111ull},
        {"M 0-3 | D 4-7", 0x44441111ull},
        {"M 0-3 | R+D 4-7", 0x66661111ull},
        {"M+R+D in every wave (in sequence per step)", 0x77777777ull},
        {"M+R in every wave", 0x33333333ull},
        {"M all | nothing else (ref)", 0x11111111ull},
        {"M16 (16x16x32) in all 8 waves", 0x88888888ull},
        {"M16 in waves 0-3", 0x00008888ull},
        {"M16+R+D in every wave", 0xeeeeeeeeull},
        {"M16 0-3 | R+D 4-7", 0x66668888ull},
    };
    for (int zero : {0, 1, 2, 4, 6, 8, 10}) {
        printf("---- operands: %s (%d), %d steps, grid 256 x 512 threads\n", zero == 0 ? "random" : zero == 1 ? "ZERO" : "random, top mantissa bits kept = code - 2", zero, steps);
        for (auto& c : cfg) {
            if (zero >= 2 && c.roles != 0x11111111ull && c.roles != 0x88888888ull) continue;
            const float us = run(256, 512, steps, c.roles, src, n, sink, zero);
            printf("%-48s %9.1f us   (%.0f cycles/step at 2.4 GHz)\n", c.name, us, us * 2400.0 / steps);
        }
    }
    return 0;
}
