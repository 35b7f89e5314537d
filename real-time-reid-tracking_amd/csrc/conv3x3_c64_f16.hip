// Layer-1 convolutions of the fp16 embed path: 3x3 / stride 1 / pad 1, 64 -> 64 channels on 64 x 32 maps (four per forward,
// 15 % of the network's MACs, SERes18_IBN.py:97-128 via SURVEY.md Appendix A).
//
// With Cout = 64 an implicit-GEMM tile is 256 x 64: every weight fragment read from LDS feeds one MFMA, every K-tile needs a
// block barrier, and the A gather re-reads the input nine times (measured 0.50 PFLOP/s, gemm_f16.hip / conv3x3_f16.hip).
// Here the whole 64 x 576 weight matrix is REGISTER resident for the life of the block: it is the MFMA A operand (rows =
// output channels), each wave owns 32 channels = 36 fragments = 144 VGPRs, loaded once and reused for every image the
// (persistent) block processes.  LDS then only holds activations:
//   * a ring of 18 input rows (34 pixels x 128 B, zero pad columns, XOR-swizzled 16-byte channel groups) filled by LDS-DMA
//     eight rows ahead; the B operand of tap (r, s) for output pixel x is row y+r-1, pixel x+s - nine shifted reads of the
//     same ring, no re-gather, no weight traffic, no barrier inside a 256-pixel step;
//   * a 256 x 64 f16 staging tile: with channels on the accumulator rows a lane owns four consecutive channels per register
//     quad, so the tile is written with packed 8-byte stores and read back as whole 16-byte channel octets, which is also where
//     the BN shift, the residual (a second staging tile, LDS-DMA during the MFMA phase), the ReLU and the per-(image, channel)
//     sum / sum-of-squares for InstanceNorm and SE pooling are applied - all on coalesced 16-byte accesses.
// The BN scale is folded into the f16 weights by the caller (max |w| rounding error 2^-11 either way).
#include "reid_internal.h"

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef f16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(uintptr_t)(p))

constexpr int MW = 32, MH = 64, MC = 64;        // map width / height / channels
constexpr int HROW_B = (MW + 2) * 128;          // ring row: 34 pixels x 64 channels f16
constexpr int RING_ROWS = 18;                   // rows 8s-1 .. 8s+16: the step's ten rows and the next eight
constexpr int RING_B = RING_ROWS * HROW_B;      // 78 336
constexpr int TILE_B = 256 * 128;               // 32 768
constexpr int STEPS = MH / 8;

__device__ __forceinline__ int ring_slot(int row) { return (row + 1) % RING_ROWS; }   // row >= -1

struct C64Params {
    const f16* in;         // [n][64][32][64]
    const f16* w;          // [64][9][64] (BN scale folded in)
    const float* shift;    // [64] or null
    const f16* residual;   // like out, or null
    float* stats;          // [n][64][2] per-image sum / sum of squares of the stored values, or null
    f16* out;
    const f16* zero_page;
    int n, relu;
    // fused SE tail (SEBasicBlock.forward, SERes18_IBN.py:120-128): when se_w1 is set, `out` receives the BLOCK output
    // relu(gate * y + shortcut) with gate = sigmoid(W2 relu(W1 avgpool(y))), y = this convolution's result (kept in the
    // per-block scratch image `y_scratch`, 2048 x 64 f16, which stays in L2) and shortcut = `residual`
    const float* se_w1;    // [8][64]
    const float* se_w2t;   // [8][64] (fc2 transposed)
    f16* y_scratch;        // [gridDim.x][2048][64]
};

// 512 threads = 8 waves = two per SIMD, 256 VGPRs each.  Wave (chh, pg) owns output channels chh*32 .. +32 (its 36 weight
// fragments = 144 VGPRs) and rows 2pg, 2pg+1 of the 8-row step.  Activation fragments are fetched two k-steps ahead of
// the MFMAs that use them (explicit rotation over three register sets); the DMA pieces of the next step's rows and of
// this step's residual tile are issued between k-steps.  Three block barriers per 256-pixel step.
template <bool HAS_RES, bool HAS_SHIFT>
__global__ __launch_bounds__(512) void conv3x3_c64_f16_kernel(const C64Params p) {
    __shared__ __attribute__((aligned(16))) char lds[RING_B + 2 * TILE_B + 256];
    char* ring = lds;
    char* tile = lds + RING_B;
    char* rtile = tile + TILE_B;                      // residual of the step, [256 px][64 ch] linear
    float* sh_lds = (float*)(rtile + TILE_B);         // BN shift
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int chh = wave & 1, pg = wave >> 1;

    half8 wf[36];
#pragma unroll
    for (int ks = 0; ks < 36; ++ks) wf[ks] = *(const half8*)(p.w + (chh * 32 + li) * 576 + (ks >> 2) * 64 + (ks & 3) * 16 + lh * 8);

    // zero pad columns (pixels 0 and 33) of every ring row: never written again
    if (tid < RING_ROWS * 16) {
        half8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (f16)0.f;
        *(half8*)(ring + (tid >> 4) * HROW_B + ((tid >> 3) & 1) * 33 * 128 + (tid & 7) * 16) = z;
    }
    if (tid < 64) sh_lds[tid] = HAS_SHIFT ? p.shift[tid] : 0.f;
    const int c8 = tid & 7;   // channel octet of this thread in the store pass
    const float lo = p.relu ? 0.f : -INFINITY;
    int voff[3][4];           // halo pixel li+sx, channel group (kk, lh) at its swizzled slot
#pragma unroll
    for (int sx = 0; sx < 3; ++sx)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) voff[sx][kk] = (li + sx) * 128 + (((kk * 2 + lh) ^ (((li + sx) >> 1) & 7)) * 16);

    for (int img = blockIdx.x; img < p.n; img += gridDim.x) {
        const f16* in_img = p.in + (long long)img * MH * MW * MC;
        // one DMA piece = 8 pixels x 128 B of one row: lane -> pixel g*8 + lane/8, slot lane%8 holds channel octet slot ^ key
        auto issue_piece = [&](int row, int g) {
            const int x = g * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (((x + 1) >> 1) & 7);
            const f16* src = (unsigned)row < (unsigned)MH ? in_img + ((long long)row * MW + x) * MC + c * 8 : p.zero_page;
            __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(ring + ring_slot(row) * HROW_B + 128 + g * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int i = 0; i < 5; ++i) {   // rows -1 .. 8
            const int q = wave * 5 + i;
            issue_piece((q >> 2) - 1, q & 3);
        }
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        for (int s = 0; s < STEPS; ++s) {
            const long long step_off = ((long long)img * MH + 8 * s) * MW * MC;
            // DMA work of this wave during the step: 4 pieces of the next rows (8s+9 .. 8s+16; row 64 = zero page) and
            // 4 pieces (1 KiB each, contiguous) of this step's residual tile
            auto issue_side = [&](int i) {
                if (i < 4) {
                    if (s + 1 < STEPS) {
                        const int q = wave * 4 + i;
                        issue_piece(8 * s + 9 + (q >> 2), q & 3);
                    }
                } else if (HAS_RES) {
                    const int q = wave * 4 + (i - 4);
                    __builtin_amdgcn_global_load_lds(GPTR(p.residual + step_off + (long long)q * 512 + lane * 8),
                                                     LPTR(rtile + q * 1024), 16, 0, 0);
                }
            };
            f32x16 acc[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
            // fragment of k-step ks = (tap (r, sx), kk) for row 2pg+a of the step: wave-uniform row base (kept opaque so that
            // the 72 addresses are formed one add at a time next to their reads, not hoisted into 72 live registers) +
            // the persistent per-lane offset voff[sx][kk]
            auto frag = [&](int ks, int a) -> half8 {
                const int tap = ks >> 2, kk = ks & 3;
                const int r = tap / 3, sx = tap - r * 3;
                int rb = ring_slot(8 * s + 2 * pg + a + r - 1) * HROW_B;
                asm volatile("" : "+s"(rb));
                return *(const half8*)(ring + rb + voff[sx][kk]);
            };
            half8 fb[3][2];
            fb[0][0] = frag(0, 0); fb[0][1] = frag(0, 1);
            fb[1][0] = frag(1, 0); fb[1][1] = frag(1, 1);
#pragma unroll
            for (int ks = 0; ks < 36; ++ks) {
                if (ks + 2 < 36) {
                    fb[(ks + 2) % 3][0] = frag(ks + 2, 0);
                    fb[(ks + 2) % 3][1] = frag(ks + 2, 1);
                }
                if ((ks & 3) == 1 && (ks >> 2) < 8) issue_side(ks >> 2);
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], fb[ks % 3][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], fb[ks % 3][1], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();   // the ring rows of this step are consumed
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int pl = (2 * pg + a) * 32 + li;
                char* t = tile + pl * 128 + lh * 8;
                const int key = (pl >> 1) & 7;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (f16)acc[a][q * 4 + e];
                    *(half4*)(t + (((chh * 4 + q) ^ key) * 16)) = v;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next rows and the residual tile have landed
            __syncthreads();
            float sh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sh[e] = sh_lds[c8 * 8 + e];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 512 * i;
                const int pl = idx >> 3;
                const half8 v = *(const half8*)(tile + pl * 128 + ((c8 ^ ((pl >> 1) & 7)) * 16));
                half8 rr;
                if (HAS_RES) rr = *(const half8*)(rtile + idx * 16);
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)v[e];
                    if (HAS_SHIFT) f += sh[e];
                    if (HAS_RES) f += (float)rr[e];
                    if (HAS_SHIFT || HAS_RES) f = fmaxf(f, lo);
                    s1[e] += f;
                    s2[e] += f * f;
                    o[e] = (f16)f;
                }
                if (p.se_w1) *(half8*)(p.y_scratch + ((long long)blockIdx.x * MH + 8 * s) * MW * MC + (long long)idx * 8) = o;
                else *(half8*)(p.out + step_off + (long long)idx * 8) = (HAS_SHIFT || HAS_RES) ? o : v;
            }
            __syncthreads();   // staging tiles are free again (the next step's residual DMA may start)
        }
        if (p.stats || p.se_w1) {   // 64 threads share a channel octet: fixed-order reduction through the staging tiles
            float* red = (float*)tile;   // [512][16] floats = both staging tiles
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[tid * 16 + e] = s1[e];
                red[tid * 16 + 8 + e] = s2[e];
            }
            __syncthreads();
            float tot = 0.f;
            if (tid < 128) {
                const int oc = tid >> 4, j = tid & 15;
                for (int m = 0; m < 64; ++m) tot += red[(oc + 8 * m) * 16 + j];
                if (p.stats) p.stats[((long long)img * MC + oc * 8 + (j & 7)) * 2 + (j >> 3)] = tot;
            }
            __syncthreads();
            if (p.se_w1) {
                // SE gate of this image: pooled mean -> 64 -> 8 (ReLU) -> 64 (sigmoid); then the block output over the
                // eight steps again: y from the scratch image this block just wrote (L2), shortcut from the residual input
                float* pooled = (float*)tile;            // [64]
                float* hid = pooled + 64;                // [8]
                float* gate = pooled + 128;              // [64]
                if (tid < 128 && (tid & 15) < 8) pooled[(tid >> 4) * 8 + (tid & 7)] = tot / (float)(MH * MW);
                __syncthreads();
                if (tid < 8 * 64) {   // wave m: hidden unit m
                    const int m = tid >> 6;
                    float a = p.se_w1[m * 64 + lane] * pooled[lane];
                    for (int o = 32; o; o >>= 1) a += __shfl_xor(a, o);
                    if (lane == 0) hid[m] = fmaxf(a, 0.f);
                }
                __syncthreads();
                if (tid < 64) {
                    float a = 0.f;
#pragma unroll
                    for (int m = 0; m < 8; ++m) a += p.se_w2t[m * 64 + tid] * hid[m];
                    gate[tid] = 1.0f / (1.0f + expf(-a));
                }
                __syncthreads();
                float g[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = gate[c8 * 8 + e];
                const f16* ys = p.y_scratch + (long long)blockIdx.x * MH * MW * MC;
                const long long img_off = (long long)img * MH * MW * MC;
                for (int s = 0; s < STEPS; ++s) {
                    half8 yv[4], rv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const long long o = (long long)s * 8 * MW * MC + (long long)(tid + 512 * i) * 8;
                        yv[i] = *(const half8*)(ys + o);
                        rv[i] = *(const half8*)(p.residual + img_off + o);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        half8 o8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o8[e] = (f16)fmaxf(g[e] * (float)yv[i][e] + (float)rv[i][e], 0.f);
                        *(half8*)(p.out + img_off + (long long)s * 8 * MW * MC + (long long)(tid + 512 * i) * 8) = o8;
                    }
                }
                __syncthreads();   // the gate lives in the staging tile the next image will overwrite
            }
        }
    }
}

// w [Cout][K] fp32 (K = 9 * Cin, KRSC) x scale[Cout] -> f16
__global__ void scale_rows_f16_kernel(const float* __restrict__ w, const float* __restrict__ scale, int k, long long total,
                                      f16* __restrict__ out) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < total) out[i] = (f16)(w[i] * scale[i / k]);
}

}  // namespace

bool conv3x3_c64_f16_supported(int H, int W, int Cin, int Cout, int R, int S, int stride, int pad) {
    return H == MH && W == MW && Cin == MC && Cout == MC && R == 3 && S == 3 && stride == 1 && pad == 1;
}

int launch_scale_rows_f16(reid_ctx* ctx, const float* w, const float* scale, int rows, int k, f16* out) {
    const long long total = (long long)rows * k;
    hipLaunchKernelGGL(scale_rows_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, w, scale, k,
                       total, out);
    LAUNCH_CHECK();
    return REID_OK;
}

// stats (if given) are per IMAGE: [n][64][2] - the finalize kernels take tiles = 1 for this producer
int launch_conv3x3_c64_f16(reid_ctx* ctx, const f16* in, int n, const f16* w_scaled, const float* shift, const f16* residual,
                           int relu, float* stats, f16* out, const f16* zero_page, const float* se_w1, const float* se_w2t) {
    C64Params p;
    p.in = in; p.w = w_scaled; p.shift = shift; p.residual = residual; p.stats = stats; p.out = out;
    p.zero_page = zero_page; p.n = n; p.relu = relu;
    p.se_w1 = se_w1; p.se_w2t = se_w2t; p.y_scratch = nullptr;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const int grid = n < cus ? n : cus;
    if (se_w1) {
        ARG_CHECK(se_w2t && residual && shift);
        REID_TRY(ctx_ws(ctx, "c64.y", (size_t)grid * MH * MW * MC * 2, (void**)&p.y_scratch));
    }
    const double flops = 2.0 * n * MH * MW * MC * 9.0 * MC;
    const double bytes = (double)n * MH * MW * MC * 2.0 * (residual ? 3.0 : 2.0);
    prof_begin(ctx, REID_K_CONV_GEMM, flops, bytes);
    if (residual && shift) hipLaunchKernelGGL((conv3x3_c64_f16_kernel<true, true>), dim3(grid), dim3(512), 0, ctx->stream, p);
    else if (residual) hipLaunchKernelGGL((conv3x3_c64_f16_kernel<true, false>), dim3(grid), dim3(512), 0, ctx->stream, p);
    else if (shift) hipLaunchKernelGGL((conv3x3_c64_f16_kernel<false, true>), dim3(grid), dim3(512), 0, ctx->stream, p);
    else hipLaunchKernelGGL((conv3x3_c64_f16_kernel<false, false>), dim3(grid), dim3(512), 0, ctx->stream, p);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
