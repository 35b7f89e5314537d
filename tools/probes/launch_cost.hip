// What does a workgroup of the halo convolution's shape cost when it does NOTHING?  (round 5: the round-4 ablation listed an
// "empty kernel" at 20 % of the layer-4 launch.)  hipcc --offload-arch=gfx950 -O3 tools/probes/launch_cost.hip -o tools/probes/launch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int THREADS, int LDS_KB, int VREGS>
__global__ __launch_bounds__(THREADS) void empty_kernel(int* out, int flag) {
    __shared__ char lds[LDS_KB * 1024];
    if (flag) {   // never taken: keeps the LDS allocation and a high register in the kernel's footprint
        lds[threadIdx.x] = 1;
        __syncthreads();
        if constexpr (VREGS == 144) asm volatile("v_mov_b32 v143, 0" ::: "v143", "memory");
        if constexpr (VREGS == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127", "memory");
        out[blockIdx.x] = lds[threadIdx.x ^ 1];
    }
}

template <int THREADS, int LDS_KB, int VREGS>
float time_it(int grid, int iters, int* d) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((empty_kernel<THREADS, LDS_KB, VREGS>), dim3(grid), dim3(THREADS), 0, 0, d, 0);
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((empty_kernel<THREADS, LDS_KB, VREGS>), dim3(grid), dim3(THREADS), 0, 0, d, 0);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1000.f;
}

int main() {
    int* d;
    CK(hipMalloc(&d, 1 << 20));
    for (int grid : {256, 2048, 8192}) {
        printf("grid %5d: 768 thr 138 KB 144 vgpr %8.1f us | 768 thr 138 KB 32 vgpr %8.1f | 768 thr 8 KB 144 vgpr %8.1f | 512 thr 138 KB 144 vgpr %8.1f | 512 thr 69 KB 128 vgpr %8.1f | 256 thr 8 KB 32 vgpr %8.1f\n",
               grid, time_it<768, 138, 144>(grid, 20, d), time_it<768, 138, 32>(grid, 20, d), time_it<768, 8, 144>(grid, 20, d),
               time_it<512, 138, 144>(grid, 20, d), time_it<512, 69, 128>(grid, 20, d), time_it<256, 8, 32>(grid, 20, d));
    }
    return 0;
}
