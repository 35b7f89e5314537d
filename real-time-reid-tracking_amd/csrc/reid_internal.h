// Internal declarations shared by the HIP translation units of libreid_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string>
#include <map>
#include <tuple>
#include <vector>
#include "../../include/reid_hip.h"
struct reid_ctx;

void reid_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            reid_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return REID_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define REID_TRY(expr)            \
    do {                          \
        int _s = (expr);          \
        if (_s != REID_OK) return _s; \
    } while (0)

#define ARG_CHECK(cond)                                                         \
    do {                                                                        \
        if (!(cond)) {                                                          \
            reid_set_error("%s:%d bad argument: %s", __FILE__, __LINE__, #cond); \
            return REID_ERR_ARG;                                                \
        }                                                                       \
    } while (0)

// ------------------------------------------------------------------------------------------------
// GEMM  C[M][N] = A'[M][K] . B[N][K]^T  on v_mfma_f32_32x32x2_f32, A' produced by one of the loaders.
// ------------------------------------------------------------------------------------------------
enum AMode {
    A_DENSE = 0,     // A[M][lda], K contiguous, K % 4 == 0, 16-byte aligned rows
    A_IM2COL = 1,    // NHWC fp32 activations, k = (r*S+s)*Cin + c, Cin % 32 == 0
    A_STEM_F32 = 2,  // NHWC fp32 [N][H][W][3], 7x7 s2 p3, k = r*24 + (s*3+c), 8 x 24 = 192 (zero padded)
    A_STEM_U8 = 3    // same from uint8 crops with (v/255-0.5)/0.5 fused into the loader
};
enum EpiMode {
    E_CONV = 0,  // optional col scale/shift, residual, relu, per-(m-tile, col) sum/sumsq partials
    E_DIST = 1,  // distance epilogue from row/col squared norms
    E_BIAS = 2   // optional bias[n], GELU, residual, ConvTranspose parity scatter
};

struct GemmParams {
    const void* A;
    long long lda;
    // im2col geometry
    int H, W, Cin, Ho, Wo, R, S, stride, pad_y, pad_x;
    const float* a_scale;  // [img][Cin] input transform x' = relu?(x*scale+shift) applied to in-bounds pixels
    const float* a_shift;
    int a_relu;
    const float* B;        // [N][ldb]
    long long ldb;
    int M, N, K;
    float* C;
    long long ldc;
    const float* col_scale;
    const float* col_shift;
    const float* residual;
    int relu;
    int relu_from;         // conv_f32.hip: ReLU applies to columns >= relu_from (0 = all; IBN: BatchNorm half only)
    float* stats;          // [M/128][N][2] or null
    // E_BIAS extras: act 1 = exact (erf) GELU after the bias; residual is added after the activation.
    // scat_h > 0: row m = (img, j, i) of a scat_h x scat_w grid is written to pixel (2j+scat_py, 2i+scat_px) of the
    // 2x up-sampled grid (one output parity of a ConvTranspose2d(4, 2, 1)); residual uses the same index.
    int act, scat_h, scat_w, scat_py, scat_px;
    int split_k;           // conv_f32_dma (small launches): blocks per output tile, fp32 partial tiles, per-tile arrival counters
    float* splitk_ws;
    int* splitk_cnt;
    unsigned scat_mhw, scat_mw;   // ceil(2^32 / (scat_h * scat_w)), ceil(2^32 / scat_w): row -> (img, j, i) by multiply-high (set_scatter)
    // par4 != 0 (conv_f32_dma general variant): the FOUR output parities of a ConvTranspose2d(4, 2, 1) in one launch - the grid
    // is four copies of the tile grid, copy q = (py, px) uses weights B + q * par_stride, padding (1 - py, 1 - px), scatter (py, px)
    int par4;
    long long par_stride;
    unsigned long long* diag;   // diagnostic builds of conv_f32.hip only: per-wave cycle sums [block<64][8][5]
    const float* row_sq;   // E_DIST
    const float* col_sq;
    int metric;
};

// ---- distance GEMM with fused selection (dist_select.hip): the k smallest per row, no m x n matrix
constexpr int SEL_CAP = 128;    // keys per (row, segment) candidate list
constexpr int SEL_KMAX = 64;    // largest k of the fused path (a list is compacted to k once it passes 64 keys)
struct SelectParams {
    const float* A;             // x [M][lda]
    const float* B;             // y [N][ldb]
    int lda, ldb, M, N, K;      // K % 64 == 0 (zero-padded rows)
    const float* row_sq;        // |x_i|^2, |y_j|^2 (null for REID_METRIC_DOT)
    const float* col_sq;
    unsigned int* gmin;         // [M][k] group minima of a sample of y as order-preserving keys (null: start from +inf)
    int metric, k, S, index_base, exp_skip;
    unsigned long long* lists;  // scratch [M][S][SEL_CAP] (k > 1)
    int* counts;                // scratch [M][S]: lengths of the lists when the sweep ends (k > 1)
    unsigned long long* final_keys;   // scratch [M][S]: every segment's arg-min key (k = 1)
};
int select_segments(int m, int n);
bool dist_select_supported(const SelectParams& p);
int launch_dist_bound(reid_ctx* ctx, const SelectParams& p);
int launch_dist_select(reid_ctx* ctx, const SelectParams& p, float* d_D, int32_t* d_I);

// ---- fp16-input / fp32-accumulate GEMM (gemm_f16.hip)
enum A16Mode {
    A16_DENSE = 0,   // A[M][lda] f16
    A16_IM2COL = 1,  // NHWC f16 activations, Cin % 64 == 0
    A16_STEM = 2     // zero-padded NHWC4 f16 image [n][Hp][Wp][4], 7x7 s2 as k = r*32 + s*4 + c (K = 256)
};
struct Gemm16Params {
    const _Float16* A;
    long long lda;
    int H, W, Cin, Ho, Wo, R, S, stride, pad, Hp, Wp;
    const _Float16* B;  // [N][ldb]
    long long ldb;
    int M, N, K;
    _Float16* C;
    long long ldc;
    const float* col_scale;
    const float* col_shift;
    const _Float16* residual;
    int relu;
    float* stats;
    const _Float16* zero_page;
    unsigned long long* diag;   // diagnostic builds only: per-wave cycle sums [block<64][wave][4]
    // linear-layer epilogue (Swin): out = act(acc + col_shift) + res32, stored as fp32 (C32) or f16 (C); ragged M, N
    int lin, act, n_real;       // lin != 0 selects it; act 1 = erf-GELU; n_real = columns that exist (0: N)
    int asym, pad_y, pad_x;     // asym != 0: separate top / left padding of the im2col gather (ConvTranspose parities)
    int scat_h, scat_w, scat_py, scat_px;   // > 0: output row (img, j, i) -> (img, 2j+py, 2i+px) of a 2x upsampled map
    float* C32;
    const float* res32;
    // conv3x3_f16.hip split-K (small launches): blocks per output tile, fp32 partial tiles, per-tile arrival counters
    unsigned scat_mhw, scat_mw; // as GemmParams
    int par4;                   // as GemmParams.par4 (LIN im2col build): four ConvTranspose parities in one launch
    long long par_stride;
    int split_k;
    float* splitk_ws;
    int* splitk_cnt;
    // conv3x3_f16.hip, SPLIT build ("fp32-class" arithmetic on the f16 matrix pipe, reid_ctx_set_precision(ctx, 2)):
    // x = xh + xl with xh = f16(x), xl' = f16((x - xh) * 2^11); x.w ~= xh.wh + (xl'.wh + xh.wl') / 2^11.  A holds [xh | xl'] per
    // pixel (2C f16 = the bytes of fp32), the weights [wh * 2^11 | wh | wl'] per tap (Cin = 3C "virtual" channels: the third
    // third re-reads xh), so ONE fp32 accumulator collects 2^11 times the product; acc_scale = 2^-11 goes into the BN scale.
    // Output fp32 (C32), residual fp32 (res32), ReLU from column relu_from on.
    _Float16* pack16;        // SPLIT builds: columns >= pack_from leave as [yh | yl'] f16 [M][2N] here instead of fp32 in C32
    int pack_from;
    int loader_prio;         // LDS-halo kernel: loader waves at raised issue priority
    int frag_ahead;          // LDS-halo kernel: fragment reads pinned one k-step ahead of their MFMAs
    float acc_scale;
    int relu_from;
    int split_terms;            // 3 (default) or 4 (adds the xl.wl product)
    int pack_out;               // LIN staged epilogue: C is [yh | yl'] f16 [M][2 n_real] (hi tile at column n_blk, low tile n_real further)
    int a_k;                    // LIN dense build with split_terms != 0: columns of A ([xh | xl'] = 2 K); K-tile k0 reads column k0 % a_k
    int pair_early;             // PAIR loop: the step barrier in front of the step's last MFMA group (REID_SPLIT_PAIR=2) instead of behind it
    int general_epi;            // SPLIT builds: the general (predicated, 64-bit addressed) epilogue instead of the lean one (A/B)
    int ablate;                 // experiments only (reid_debug_conv_split): phases of the PAIR loop switched off, results wrong
    int* fault;                 // reid_ctx.fault (may be null): [0] raised when a value packed as [yh | yl'] lies outside f16's range
};

// |x - y|^2 from the dot product and the squared norms: ONE explicit form for every kernel that computes or bounds a distance
// (gemm_f32.hip, gemm_f32_dma.hip, dist_select.hip's bound pass and sweep).  Left as (rs + cq) - 2 * dot the multiply-add is
// contracted or not per call site by the compiler, and a bound computed one way need not bound a value computed the other way.
__device__ __forceinline__ float l2sqr_of(float dot, float rs, float cq) { return fmaf(-2.0f, dot, rs + cq); }

// precision 2 range guard (reid_ctx.fault; include/reid_hip.h, reid_ctx_set_precision): every site that writes a split operand
// [xh | xl'] keeps the largest magnitude it packed - as bits, so that NaN > inf > every finite value - and raises fault[0] when f16
// cannot hold it.  fault[1]: a non-finite embedding left the neck.
__device__ __forceinline__ unsigned range_acc(unsigned m, float v) {
    const unsigned b = __float_as_uint(v) & 0x7fffffffu;
    return b > m ? b : m;
}
__device__ __forceinline__ void range_raise(int* fault, unsigned m) {
    if (fault && m >= 0x477fe000u) fault[0] = 1;   // 65504.0f
}

// reciprocals for the scatter epilogues: floor(x / d) == umulhi(x, ceil(2^32 / d)) for x < 2^32 / d (rows of one pass: < 2^20)
template <class P>
inline void set_scatter(P& p, int scat_h, int scat_w, int py, int px) {
    p.scat_h = scat_h; p.scat_w = scat_w; p.scat_py = py; p.scat_px = px;
    p.scat_mhw = p.scat_mw = 0;
    if (scat_h > 0) {
        p.scat_mhw = (unsigned)(((1ull << 32) + (unsigned long long)(scat_h * scat_w) - 1) / (unsigned long long)(scat_h * scat_w));
        p.scat_mw = (unsigned)(((1ull << 32) + (unsigned long long)scat_w - 1) / (unsigned long long)scat_w);
    }
}

struct reid_ctx;
int launch_gemm_f16(reid_ctx* ctx, int amode, const Gemm16Params& p, int kind, double flops, double bytes);
bool conv3x3_f16_supported(const Gemm16Params& p);   // conv3x3_f16.hip: 3x3 s1 p1 with the input halo tile kept in LDS
int launch_conv3x3_f16(reid_ctx* ctx, const Gemm16Params& p, int kind, double flops, double bytes);
int launch_conv3x3_split(reid_ctx* ctx, const Gemm16Params& p, int kind, double flops, double bytes);   // SPLIT build, see Gemm16Params
bool conv3x3_x3_supported(const reid_ctx* ctx, const Gemm16Params& p);   // conv3x3_x3.hip: the same convolution as two 4-wave blocks per CU (large launches)
int launch_conv3x3_x3(reid_ctx* ctx, const Gemm16Params& p);
bool lin_x3_supported(const reid_ctx* ctx, const Gemm16Params& p);        // conv3x3_x3.hip: the dense form (Swin linears of stages 3-4, fp32-class mode)
int launch_lin_x3(reid_ctx* ctx, const Gemm16Params& p, int kind, double flops, double bytes);
bool conv_x3s_supported(const reid_ctx* ctx, const Gemm16Params& p);      // conv3x3_x3.hip: strided 3x3 / 1x1 convolutions, fp32-class mode
int launch_conv_x3s(reid_ctx* ctx, const Gemm16Params& p, int kind, double flops, double bytes);
int launch_gemm_f16_split(reid_ctx* ctx, const Gemm16Params& p, int kind, double flops, double bytes);   // SPLIT build of gemm_f16 (im2col: strided / 1x1)
bool two_linear_supported(const reid_ctx*, long long T, int C, int hid);
int launch_two_linear(reid_ctx*, const _Float16* a16, long long T, int C, int hid, const float* w1, const float* b1, const float* w2,
                      const float* b2, int act, const float* res, float* out, const float* x32 = nullptr, const float* ln_g = nullptr,
                      const float* ln_b = nullptr);
bool ln_linear_supported(const reid_ctx*, long long T, int C, int n);
int launch_ln_linear(reid_ctx*, const float* x32, const float* ln_g, const float* ln_b, long long T, int C, int n, const float* w, const float* bias,
                     float* out, int ldc);
int launch_split_pack(reid_ctx* ctx, const float* x, long long rows, int C, _Float16* out, const float* d_scale = nullptr);           // fp32 [rows][C] -> f16 [rows][2C] = [xh | xl']
int launch_split_weights(reid_ctx* ctx, const float* w, int cout, int taps, int cin, int terms, _Float16* out, const float* d_scale = nullptr);  // fp32 [cout][taps][cin] -> f16 [cout][taps][terms * cin]
// fp16 elementwise kernels (elementwise_f16.hip)
int launch_prep_u8_pad_f16(reid_ctx*, const uint8_t* crops, int n, int h, int w, int hp, int wp, _Float16* out);
int launch_prep_f32_pad_f16(reid_ctx*, const float* nhwc3, int n, int h, int w, int hp, int wp, _Float16* out);
int launch_maxpool3s2_f16(reid_ctx*, const _Float16* x, int n, int h, int w, int c, _Float16* out);
int launch_affine_relu_f16(reid_ctx*, _Float16* x, const float* a_scale, const float* a_shift, int n_img, int hw, int c);
int launch_norm_apply_f16(reid_ctx*, _Float16* x, const float* stats, int n_img, int tiles, int c, int half, int hw,
                          const float* in_gamma, const float* in_beta, const float* bn_scale, const float* bn_shift);
int launch_se_tail_f16(reid_ctx*, const float* stats, int n_img, int tiles, int c, int mid, int hw, const float* w1, const float* w2t,
                       const _Float16* y, const _Float16* sc, _Float16* out);
int launch_se_combine_f16(reid_ctx*, const _Float16* y, const _Float16* sc, const float* s, int n_img, int hw, int c,
                          _Float16* out);
int launch_gem_neck_f16(reid_ctx*, const _Float16* x, int n_img, int hw, int c, const float* p, const float* scale,
                        const float* shift, float* gem_out, float* emb);
int launch_f32_to_f16(reid_ctx*, const float* x, size_t n, _Float16* out);
int launch_stem_w16(reid_ctx*, const float* stem_w_f32, _Float16* out);
bool conv3x3_c64_f16_supported(int H, int W, int Cin, int Cout, int R, int S, int stride, int pad);
int launch_scale_rows_f16(reid_ctx*, const float* w, const float* scale, int rows, int k, _Float16* out);
int launch_conv3x3_c64_f16(reid_ctx*, const _Float16* in, int n, const _Float16* w_scaled, const float* shift,
                           const _Float16* residual, int relu, float* stats, _Float16* out, const _Float16* zero_page,
                           const float* se_w1 = nullptr, const float* se_w2t = nullptr);
int launch_stem_w16_scaled(reid_ctx*, const float* stem_w_f32, const float* scale, _Float16* out);
int launch_stem_pool_f16(reid_ctx*, const _Float16* pad_in, const uint8_t* crops_u8, int n, const _Float16* w16s, const float* shift,
                         _Float16* pooled);
int launch_conv_w16_chunked(reid_ctx*, const float* w_f32, int cout, int rs, int cin, _Float16* out);  // -> [Cout][Cin/64][RS][64]   // [64][8][24] -> [64][8][8][4]
int launch_gemm_f32(reid_ctx* ctx, int amode, int epi, const GemmParams& p, int kind, double flops, double bytes);
int launch_ta_tail(reid_ctx* ctx, const float* y, const float* sc, int n_img, int H, int W, int C, const float* wts, float* out);
int launch_ema_tail(reid_ctx* ctx, const float* y, const float* sc, int n_img, int H, int W, int C, const float* prm, float* out);
int launch_stem_split(reid_ctx* ctx, const void* x, bool is_u8, int n, const float* wgt, const float* scale, const float* shift, float* out,
                      _Float16* packed);   // precision 2: 7x7 stem + BN + max-pool on split f16 operands (stem_split.hip)
int launch_stem_f32(reid_ctx* ctx, const void* x, bool is_u8, int n, const float* wgt, const float* scale, const float* shift,
                    float* out, bool pooled);   // stem_f32.hip: 7x7 s2 conv + BN (+ MaxPool(3,2,1) on the accumulators) of the fp32 path
bool gemm_f32_dma_supported(int amode, int epi, const GemmParams& p);   // gemm_f32_dma.hip: dense GEMM with LDS-DMA staging
int launch_gemm_f32_dma(reid_ctx* ctx, int epi, const GemmParams& p, int kind, double flops, double bytes);
bool conv_f32_supported(const GemmParams& p);   // conv_f32.hip: pipelined implicit-GEMM convolution of the fp32 path (full tiles)
int launch_conv_f32(reid_ctx* ctx, const GemmParams& p, int kind, double flops, double bytes);
bool conv_f32_general_supported(const GemmParams& p);   // same kernel, general geometry + bias epilogue (Swin's convolutions)
int launch_conv_f32_general(reid_ctx* ctx, const GemmParams& p, int kind, double flops, double bytes);

// elementwise / reduction kernels (elementwise.hip)
int launch_nchw_to_nhwc3(reid_ctx*, const float* x_nchw, int n, int h, int w, float* out_nhwc);
int launch_resize_norm(reid_ctx*, const uint8_t* packed, const long long* offsets, const int* hw, int n, int H, int W,
                       int pitch, float* out_nhwc);
int launch_maxpool3s2(reid_ctx*, const float* x, int n, int h, int w, int c, float* out);
int launch_norm_finalize(reid_ctx*, const float* stats, int n_img, int tiles, int c, int half, int hw,
                         const float* in_gamma, const float* in_beta, const float* bn_scale, const float* bn_shift,
                         float* a_scale, float* a_shift);
int launch_in_apply(reid_ctx*, float* x, const float* stats, int n_img, int tiles, int c, int half, int hw, const float* in_gamma,
                    const float* in_beta);
int launch_se_finalize(reid_ctx*, const float* stats, int n_img, int tiles, int c, int mid, int hw, const float* w1,
                       const float* w2, float* s);
int tail_slices(int n_img, int hw);   // elementwise.hip: blocks per image of the fused tail kernels
int launch_se_tail(reid_ctx*, const float* stats, int n_img, int tiles, int c, int mid, int hw, const float* w1, const float* w2,
                   const float* y, const float* sc, float* out, _Float16* packed = nullptr);   // packed: also [oh | ol'] f16 [.., 2c]
int launch_se_combine(reid_ctx*, const float* y, const float* sc, const float* s, int n_img, int hw, int c, float* out);
int launch_in_apply_pack(reid_ctx*, const float* x, const float* stats, int n_img, int tiles, int c, int half, int hw,
                         const float* in_gamma, const float* in_beta, _Float16* packed,   // precision 2: IBN finish -> [xh | xl']
                         bool in_half_only = false);   // the BatchNorm half was packed by the conv epilogue already
int launch_gem_neck(reid_ctx*, const float* x, int n_img, int hw, int c, const float* p, const float* scale,
                    const float* shift, float* gem_out, float* emb);
int launch_row_sqnorm(reid_ctx*, const float* x, int m, int d, long long ld, float* out);
// selection kernels (select.hip)
int launch_argmin_rows(reid_ctx*, const float* dist, int m, int n, long long ld, int32_t* idx, float* val);
int launch_topk_rows(reid_ctx*, const float* dist, int m, int n, long long ld, int k, float* D, int32_t* I);
int launch_rank_eval(reid_ctx*, const float* score, int nq, int ng, long long ld, const long long* ql,
                     const long long* qc, const long long* gl, const long long* gc, int32_t* cmc_sum, double* ap,
                     int32_t* valid);
int launch_diou_cost(reid_ctx*, const double* tracks, int t, const double* dets, int m, double* out, int as_cost);

// ------------------------------------------------------------------------------------------------
struct ProfSlot {
    double ms = 0, flops = 0, bytes = 0;
    long long launches = 0;
};

struct Se18Block {
    int c, cin, stride, ibn, ds, mid;
    const float *conv1_w, *in_gamma, *in_beta, *bn1_scale, *bn1_shift, *conv2_w, *bn2_scale, *bn2_shift;
    const float *ds_w, *ds_scale, *ds_shift, *se_w1, *se_w2;
    const float *ta, *ema;   // sibling backbones: TripletAttention gates [3][100] / EMA parameters (attention_f32.hip)
};

struct Se18Weights {
    bool loaded = false;
    float* blob = nullptr;
    size_t n_floats = 0;
    int num_class = 0;
    int arch = 0;                 // 0 SERse18_IBN, 1 CARes18_IBN (TripletAttention), 2 EMARes18_IBN
    const float *stem_w, *stem_scale, *stem_shift;
    Se18Block blk[8];
    const float *gem_p, *neck_scale, *neck_shift, *cls_w;
    const float* cam_bias = nullptr;   // [num_cams][512] (cam.bias; optional) and the constructor's cam_factor (cam.factor)
    int num_cams = 0;
    float cam_factor = -1.0f;
    _Float16* blob16 = nullptr;   // fp16 copy of the whole blob (same element offsets) for the fp16 path
    _Float16* stem_w16 = nullptr; // [64][256] stem weights of the padded-NHWC4 formulation
    _Float16* l1_conv2_w16s[2] = {nullptr, nullptr};   // layer-1 conv2 weights x BN scale (conv3x3_c64_f16.hip)
    _Float16* stem_w16s = nullptr; // same with the folded BN scale multiplied in (fused stem + maxpool kernel)
    _Float16* zero_page = nullptr;
    float* ep = nullptr;          // [8 blocks][2][512]: conv1 epilogue scale / shift of the IBN blocks: (1, 0) on the InstanceNorm
                                  // half, the folded BatchNorm on the other (conv_f32.hip flow)
    const _Float16* h(const float* p) const { return blob16 + (p - blob); }
};

// comm.hip: the RCCL communicator of this rank.  `loop` is a TEST transport (libreid_hip_debug.so, reid_debug_comm_loopback):
// several contexts of ONE process on ONE device act as the ranks, so that the multi-rank C code (ragged gathers, index_base,
// padding rows, the merge) runs with world > 1 on a one-GPU box.  The product never sets it: reid_comm_init is RCCL only.
struct reid_comm_loop {
    virtual int allgather(int rank, const void* d_send, void* d_recv, size_t bytes, hipStream_t st) = 0;
    virtual int allreduce(int rank, double* inout, int count, int op) = 0;
    virtual void detach(int rank) = 0;
    virtual ~reid_comm_loop() {}
};
struct reid_comm {
    void* comm = nullptr;   // ncclComm_t
    int rank = 0, world = 1;
    reid_comm_loop* loop = nullptr;
};
void comm_release(reid_ctx* ctx);
// k-way merge of per-shard top-k lists [world][nq][kk] (global indices, -1 = padding) -> [nq][k] (comm.hip)
int launch_knn_merge(reid_ctx* ctx, const float* Dall, const int32_t* Iall, int world, int nq, int kk, int k, float* D, int32_t* I);

struct reid_ctx {
    int device = 0;
    reid_comm* comm = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    int chunk = 1024;        // crops per pass (reid_ctx_set_chunk).  64 until round 4: 45 k crops/s in mode 2 where 1024 gives 68 k; the parity
                             // sets give the reference's arg-min on every row at 64, 128, 256 and 1024 (tools/config1_chunk_check.py)
    int precision = 0;
    bool profile = false;
    ProfSlot prof[REID_K_COUNT];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    struct Pending { int kind; hipEvent_t a, b; double flops, bytes; };
    std::vector<Pending> pending;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // growable device workspaces, keyed by name
    std::map<std::string, std::pair<void*, size_t>> ws;
    std::map<std::string, std::pair<void*, size_t>> pinned;   // pinned host staging buffers (ctx_pinned)
    std::map<const void*, void*> split_w;   // precision 2: conv weights (fp32, in the blob) -> their [wh * 2^11 | wh | wl'] f16 form
    // Switches below: fixed defaults in the product library.  The REID_* names in their comments are the environment variables that
    // set them until round 4; since round 5 the product library reads none of them (api.hip, reid_ctx_create) and experiments move a
    // field by its NAME through libreid_hip_debug.so: reid_debug_set_switch(ctx, "split_pair", 1) (debug.hip, kSwitches).
    int bank_fast = 1;                    // d = 512 feature-bank cost on the register-tiled kernel (REID_BANK_FAST=0: generic kernel)
    Se18Weights se18;
    int last_n = 0;  // crops in the last embed chunk (for reid_debug_stage)
    int swin_last_n = 0, swin_last_tok = 0;   // images / stage-1 tokens per image of the last Swin pass (reid_debug_swin_stage)
    int debug_keep = 0;      // 0 off, 1 stage buffers + unfused kernels, 2 stage buffers + production kernels
    bool last_f16 = false;
    int f16_wide_splitk = 1;  // LDS-halo kernel, small launches with a long K loop: 128-wide tiles split four ways (REID_F16_WIDE_SPLITK=0: 64-wide)
    int f16_frag_ahead = 1;   // LDS-halo kernel: fragments of k-step kk + 1 read before the MFMAs of kk (REID_F16_FRAG_AHEAD=0: compiler's order)
    int f16_loader_prio = 1;  // loader waves of the LDS-halo kernel at s_setprio 3: 1 = SPLIT builds (mode 2: 15.25 -> 15.10 ms per 1024 crops;
                              // the plain f16 build loses 0.6 %), 2 = all, 0 = none (REID_F16_LOADER_PRIO)
    int f16_loader_waves = 1; // LDS-halo kernel: 8 compute + 4 dedicated loader waves (REID_F16_LOADERS)
    int f16_halo = 1;        // 3x3 stride-1 convs of the fp16 path use the LDS-halo kernel (REID_F16_HALO=0: implicit GEMM)
    int f16_se_tail = 1;     // fp16 path: SE gate + combine in one launch per block (REID_F16_SETAIL=0: se_finalize + se_combine)
    int f16_c64 = 2;         // fp16 path: layer-1 convs on the register-resident-weight kernel, 2 = with the SE tail fused
                             // into conv2 (REID_F16_C64=1: separate se_finalize / se_combine kernels, 0: implicit GEMM)
    int f16_stem_fused = 2;  // fp16 path: stem conv + BN + maxpool as one kernel, 2 = fed with the uint8 crops directly
                             // (REID_F16_STEMPOOL=1: from the padded f16 image, 0: GEMM + pool kernels)
    int f32_conv = 1;        // fp32 path (REID_F32_CONV): 1 = conv_f32.hip LDS-DMA kernel, norms in the producer's epilogue / in_apply;
                             // 2 = conv_f32.hip register-staged kernel, norm in the loader; 0 = gemm_f32_kernel<A_IM2COL> (round 1)
    int f32_split_k = 1;     // fp32 conv: split the K-tiles over 2-4 blocks per output tile when a launch has <= 256 tiles (REID_F32_SPLITK)
    int lin_x3 = 1;          // Swin, fp32-class mode: linears with N % 128 == 0 on conv3x3_x3.hip's dense kernel (0: gemm_f16.hip's linear build)
    int f32_dist_bk16 = 1;   // distance matrix, 128-wide tiles: K-tiles of 16 -> three blocks per CU (gemm_f32_dma.hip); 0 = K-tiles of 32, two blocks
    int pack_epilogue = 1;   // precision 2: conv1 epilogues write [yh | yl'] for conv2 themselves (REID_PACK_EPILOGUE=0: fp32 + pack passes)
    int split_lean_epi = 1;  // precision 2: buffer-instruction epilogue of the SPLIT convolution builds (REID_SPLIT_LEAN=0: general loop)
    int split_x3 = 3;        // precision 2, large launches: conv3x3_x3.hip instead of conv3x3_f16.hip's 12-wave kernel: 3 (default) = on
                             // v_mfma_f32_16x16x32_f16, two blocks per CU for the 128-wide tiles (layers 2-4) and FOUR for the 64-wide ones
                             // (layer 1); 2 = the 128-wide tiles only; 1 = the first form on 32x32x16; 0 = off
    int split_gemm_min_tiles = 128;   // precision 2: strided / 1x1 convolutions take the SPLIT build of gemm_f16.hip from this many 256 x 128 tiles on (below: the exact-fp32 split-K kernel)
    int chain = 0;           // EXPERIMENTS=1 builds only: precision 2, batches of up to 64 crops, layer 4 as ONE chain launch (conv3x3_x3.hip chain_kernel;
                             // 536 us against 255 us for the six launches it replaces: DESIGN.md section 8); bit 2 = timing variant (wrong results)
    int x3_sk_cap = 0;       // experiments: upper bound of the split-K factor of conv3x3_x3.hip's small launches (0 = the heuristic's)
    int split_x3_small = 2;  // ... and smaller launches: 2 (default) = where they measured faster than conv3x3_f16.hip's 12-wave kernel (conv3x3_x3_supported),
                             // 1 = every launch, 0 = none; K split over up to 8 blocks per tile (x3m16_tail: reduce-scatter)
    int x3s_sk_cap = 8;      // conv_x3s_kernel: most blocks per output tile
    int conv_x3s = 1;        // conv3x3_x3.hip conv_x3s_kernel for the strided 3x3 / 1x1 convolutions of the fp32-class mode: 1 = wherever gemm_f16.hip's SPLIT build served (launches of >= ~100 tiles), 2 = also the small launches that run in exact fp32, 0 = off
    int x3_l4_narrow_nmt = 31;  // conv3x3_x3.hip x3_wide_tiles: layer 4 on 64-wide tiles where its 128-wide launch fills the chip unevenly (0: never; split launches up to this many tile rows)
    int x3_narrow = 1;       // conv3x3_x3.hip, 64-wide tiles (four blocks per CU) beyond layer 1: bit 0 = the 16-wide maps (layer 2: 14.03 -> 13.87 ms per
                             // 1024-crop pass; default), bit 1 = the 8-wide ones (layers 3-4: 14.03 -> 14.70, off)
    int x3_unroll = 3;       // conv3x3_x3.hip: the form with a chunk's 27 steps unrolled (addresses, DMA offsets made once, waits immediates): 1 = the
                             // 128-wide tiles, 3 (default) = and the 64-wide ones at THREE blocks per CU (134-148 registers; at four they spill:
                             // 12.59 / 12.32 / 14.05 ms per 1024-crop pass for 1 / 3 / 4), 0 = the looped kernels
    int x3_ablate = 0;       // timing experiments on conv3x3_x3.hip (debug switch; WRONG results while set)
    int split_x3_min_blocks = 512;   // ... from this many blocks on (two for every CU)
    int split_pair = 0;      // precision 2, 128-wide halo tiles (REID_SPLIT_PAIR): 0 = three passes over the virtual channels (default),
                             // 1 / 2 = the operand-sharing PAIR order of conv3x3_f16.hip (barrier behind / in front of a step's last
                             // MFMA group).  Measured at 1024 crops per pass: +1.1 % / -0.9 %; it changes the summation order, and the
                             // config-1 noise set has a row whose top-2 gap (2.7e-7) any reordering can flip - not worth 1 %
    int stem_split = 1;      // precision 2: the 7x7 stem on split f16 operands (REID_STEM_SPLIT=0: the fp32-pipe stem + split_pack)
    int f32_stem_pool = 1;   // fp32 path: MaxPool(3,2,1) on the stem kernel's accumulators (REID_F32_STEMPOOL=0: separate kernel)
    int split_terms = 3;     // precision 2: f16 products per multiply (REID_SPLIT_TERMS=4 adds the low x low product)
    int knn_wide = 1;        // large k-NN searches: candidates on the f16 matrix pipe + exact fp32 refinement (knn_wide.hip; REID_KNN_WIDE=0:
                             // the fused fp32 search for every size)
    long long knn_wide_min = 1ll << 25;   // query x gallery pairs from which the wide path is taken (REID_KNN_WIDE_MIN): Market-size (3368 x 15913) 0.68 -> 0.41 ms
    int knn_wide_force = 0;  // tests (reid_debug_knn_wide): every row whose index is a multiple of it takes the exact-row fallback
    int select_exp = 0;      // experiments only (reid_debug_select_exp, debug.hip): 1 / 2 skip phases of the fused selection (results
                             // are then incomplete), 4 prints candidate-list statistics
    int select_two_pass = 0; // REID_SELECT_TWO_PASS=1: arg-min / k-NN through the full distance matrix (A/B against dist_select.hip)
    int swin_stop = -1;      // diagnostics (REID_SWIN_STOP = block * 10 + phase): skip the rest of the Swin blocks after that point
    int swin_fold = 1;       // Swin, fp16-storage mode: to_out and post_proj folded into one Linear (REID_SWIN_FOLD=0: two launches)
    int swin_two_linear = 1; // Swin, fp32-class mode, C = 96: to_out -> post_proj and fc1 -> GELU -> fc2 as one launch each, the hidden
                             // values in registers (two_linear_f16.hip; REID_SWIN_TWO_LINEAR=0: two launches through gemm_f16.hip)
    int swin_attn_split = 0; // REID_SWIN_ATTN_SPLIT=1: Swin window attention of the fp32-class mode on the matrix cores, split operands, instead
                             // of the exact-fp32 VALU kernel (measured: 14.0 k against 14.7 k img/s - see window_attn_mfma_split_kernel)
    int two_linear_ablate = 0;   // reid_debug_two_linear_ablate (timing experiments: WRONG results): 1 = no weight refills, 2 = no barriers
    int swin_attn_mfma = 1;  // Swin window attention (REID_SWIN_ATTN): 1 = matrix cores in fp16-storage mode, VALU kernel in exact fp32
                             // (v_mfma_f32_32x32x2_f32 runs at the fp32 VALU rate: no gain); 2 = matrix cores in both; 0 = VALU in both
    int f16_split_k = 1;     // LDS-halo kernel: split the input channels over 2-4 blocks per tile when a launch has < 128 tiles (REID_F16_SPLITK)
    int swin_chunk_cap = 1024;   // images per Swin pass at most (REID_SWIN_CHUNK_MAX lowers it, read at context creation)
    int f16_lin_256 = 1;     // fp32-class Swin linears / trunk convolutions with N % 256 == 0 and K >= 1152 on 256 x 256 tiles, BK 64 (REID_F16_LIN_256=0:
                             // 256 x 128, BK 32).  Same K order per output: bit-identical; 16.11 -> 16.25 k img/s
    int f16_cfg = 0;         // fp16 GEMM tile/ring override: BN*1000 + BK*10 + NST, 0 = heuristic (REID_F16_CFG)
    int frame_m[2] = {0, 0};                        // frame pipeline (bank.hip): detections / device embeddings per frame slot
    float* frame_emb[2] = {nullptr, nullptr};
    int frame_pending[2] = {0, 0}, frame_has[2] = {0, 0};
    size_t frame_tm[2] = {0, 0};        // elements of the cost stage's output (sum over the camera groups of tracks x detections)
    hipEvent_t frame_ev[2] = {nullptr, nullptr};
    hipStream_t match_stream = nullptr;  // reid_frame_match_stream(ctx, 1): cost / update stages of the frame pipeline on a stream of their own, so that a
                                         // group of look-ahead frames walks its serial cost -> assign -> update chain beside the next group's forward
    int match_async = 0;
    hipEvent_t join_ev = nullptr;        // orders the bank's other entry points between the two streams
    hipEvent_t fwd_ev[2] = {nullptr, nullptr}, match_ev[2] = {nullptr, nullptr};   // slot's forward queued / its last reader on the match stream queued
    hipStream_t copy_stream = nullptr;   // uploads of the frame pipeline (beside the kernels of the previous frame)
    hipEvent_t copy_ev = nullptr;
    std::vector<hipEvent_t> pipe_ev;     // host_passes (api.hip): "pass k uploaded" / "pass k computed" events of the host-in / host-out entry points
    int host_pipeline = 1;               // host entry points with more than one pass: pass k + 1's upload and pass k - 1's download on the copy stream
                                         // under pass k's kernels (0: whole batch up, compute, whole result down - the form of rounds 1-5)
    int side_copy = 1;                   // REID_SIDE_COPY=0: uploads in the compute stream
    std::vector<int32_t> side_idx;       // reid_ctx_set_side_index: camera / view index per image of the following embed call(s)
    size_t side_cursor = 0;              // how many of them the passes so far have consumed
    const char* frame_out[2] = {nullptr, nullptr};
    float* stage_ptr[11] = {nullptr};
    unsigned long long* conv_diag = nullptr;   // experiments (debug.hip): stamps of the loader-wave conv kernel
    // precision 2 guards (include/reid_hip.h, reid_ctx_set_precision): the first conv / linear weight of the loaded checkpoint that
    // cannot be split ([wh 2^11 | wh | wl'] needs |w| 2^11 < 65504), empty when all can ...
    std::string split_bad_se18, split_bad_swin;
    // ... and a sticky fault word in pinned host memory that kernels raise: REID_FAULT_RANGE = an activation outside f16's range
    // reached a split ([xh | xl']) site, REID_FAULT_NONFINITE = a non-finite embedding left the neck.  Every entry point returns
    // REID_ERR_STATE while it is set (ctx_fault_status); reid_ctx_clear_fault resets it.
    int* fault = nullptr;
};
enum { REID_FAULT_RANGE = 1, REID_FAULT_NONFINITE = 2 };
int ctx_fault_status(reid_ctx* ctx);
// knn_wide.hip: brute-force k-NN for large problems (candidates in fp32-class arithmetic, exact fp32 refinement)
bool knn_wide_eligible(reid_ctx* ctx, int nq, int nb, int d, int k);
int knn_wide_dev(reid_ctx* ctx, const float* xp, int nq, const float* yp, int nb, int ld, const float* rs, const float* cq, int k, float* d_D,
                 int32_t* d_I);   // api.hip: REID_OK, or REID_ERR_STATE + message when the fault word is set

// precision 2: largest |w| of the named tensors against the bound their split form allows; returns "" or "name (max |w| = v)"
inline std::string split_range_violation(const float* blob, const std::map<std::string, std::pair<size_t, size_t>>& tab,
                                         const std::vector<std::pair<std::string, float>>& suffix_bound) {
    for (const auto& kv : tab) {
        for (const auto& sb : suffix_bound) {
            const std::string& suf = sb.first;
            if (kv.first.size() < suf.size() || kv.first.compare(kv.first.size() - suf.size(), suf.size(), suf) != 0) continue;
            float mx = 0.f;
            const float* w = blob + kv.second.first;
            for (size_t i = 0; i < kv.second.second; ++i) {
                const float a = w[i] < 0 ? -w[i] : w[i];
                if (!(a <= mx)) mx = a;        // NaN counts as a violation
            }
            if (!(mx < sb.second)) {
                char buf[160];
                snprintf(buf, sizeof(buf), "%s (max |w| = %g, limit %g)", kv.first.c_str(), (double)mx, (double)sb.second);
                return buf;
            }
            break;
        }
    }
    return std::string();
}

// ---- chain kernels (conv3x3_x3.hip, round 6): ONE persistent launch runs all the 3x3 stride-1 convolutions, InstanceNorm finishes
// and SE tails of a ResNet layer for a small batch.  Work items are claimed in list order with one atomic; an item waits only for
// counters of ITS OWN image(s) (crops are independent in eval mode: SERes18_IBN.py:32-41,88-93,120-128), never for a grid barrier.
struct ChainElem {             // an InstanceNorm finish + pack (kind 1) or an SE gate + tail (kind 2) over the images of the batch
    const float* x;            // kind 1: conv1 output fp32 [n][hw][c] (InstanceNorm half raw); kind 2: conv2 output y fp32
    const float* stats;        // per-128-row column sums [n * tiles][c][2] of the producing convolution
    const float* sc;           // kind 2: shortcut fp32 [n][hw][c]
    const float *g, *b;        // kind 1: IN gamma / beta [half]; kind 2: SE fc1 [mid][c] / fc2^T [mid][c]
    float* out;                // kind 2: block output fp32 (may be null)
    _Float16* packed;          // [n][hw][2c] as [xh | xl'] (kind 1: the InstanceNorm half's columns only; kind 2: may be null)
    int c, half, hw, tiles, mid, slices;
};
struct ChainStage {
    int kind;                  // 0 convolution, 1 InstanceNorm finish + pack, 2 SE tail
    int idx;                   // index into ChainParams::conv / ::el
    int first, items;          // range of the stage in the item list
    int dep;                   // stage this one waits for (per image), -1: its input was ready at launch
    int target;                // ... until that stage's counter of the image has reached this
    int sk;                    // convolution: blocks per output tile
};
struct ChainParams {
    int n_img, n_stages, total_items, flags;
    ChainStage st[8];
    Gemm16Params conv[4];
    ChainElem el[4];
    int* counters;             // [0] next item, [1] blocks that have left, [64 + stage * 64 + image] items of the stage finished for the image
    int* fault;
};
int launch_chain(reid_ctx* ctx, const ChainParams& cp, int W);      // conv3x3_x3.hip

// convolution launchers of the two arithmetic modes (api.hip); also used by the experiment harnesses in debug.hip
int conv_gemm(reid_ctx* ctx, int amode, const void* x, int n, int H, int W, int Cin, const float* wgt, int Cout, int R,
              int S, int stride, int pad, int Kpad, const float* a_scale, const float* a_shift, int a_relu,
              const float* col_scale, const float* col_shift, const float* residual, int relu, float* stats, float* out,
              int relu_from = 0, const _Float16* x_packed = nullptr,
              _Float16* out_packed = nullptr, int pack_from = 0, bool* packed_written = nullptr);   // x_packed (precision 2): x already as [xh | xl']
int conv_gemm16(reid_ctx* ctx, int amode, const _Float16* x, int n, int H, int W, int Cin, const _Float16* wgt, int Cout,
                int R, int S, int stride, int pad, int K, const float* col_scale, const float* col_shift,
                const _Float16* residual, int relu, float* stats, _Float16* out, int Hp = 0, int Wp = 0);

// Every entry point runs on the context's device, whatever device the calling thread had current (hipSetDevice is
// per-thread: a worker thread starts on device 0; a host application may have switched devices).  Restores on exit.
struct DeviceGuard {
    int prev = -1, dev = -1;
    explicit DeviceGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define CTX_GUARD(ctx) DeviceGuard _dev_guard((ctx)->device)
// entry points that start new work: device guard + the sticky fault word (a raised fault makes every later call fail loudly)
#define CTX_ENTER(ctx)            \
    CTX_GUARD(ctx);               \
    REID_TRY(ctx_fault_status(ctx))

void swin_release(reid_ctx* ctx);   // frees the Swin weights held for this context (swin.hip)
int ctx_ws(reid_ctx* ctx, const char* name, size_t bytes, void** out);  // grow-only named workspace
// pending side indices of the next n images (reid_ctx_set_side_index): *d_idx = device array of n entries, or nullptr when none are
// pending; entries must be < rows (the table of the loaded weights, 0 = the weights have none)
int ctx_take_side(reid_ctx* ctx, int n, int rows, const char* what, const int32_t** d_idx);
// x[img][p][c] += coeff * table[idx[img]][c] (elementwise.hip)
int launch_add_indexed_rows(reid_ctx* ctx, float* x, int n, long long hw, int C, const float* table, const int32_t* d_idx, float coeff);
int ctx_pinned(reid_ctx* ctx, const char* name, size_t bytes, void** out);   // grow-only named pinned host buffer
int ctx_pipe_events(reid_ctx* ctx, int passes);   // api.hip: the copy stream + 2 events per pass (host_passes)

// Host in -> host out in passes.  The reference's loops move every batch across PCIe (feature_extractor.py:48-53 `.to(device)` ...
// `.cpu().numpy()`, image_reid_inference.py:116-122); a caller that hands over host buffers pays that too, so the entry points hide it:
// pass k + 1's upload and pass k - 1's download are queued on the context's copy stream while pass k's kernels run on its compute
// stream (pinned sources copy asynchronously; a pageable one blocks the HOST inside hipMemcpyAsync, after pass k has been queued, so
// the device stays busy either way).  up(i, m, s) queues the upload of items [i, i + m) on stream s, run(i, m) the kernels on
// ctx->stream, down(i, m, s) the download.  The passes are the same passes, in the same order, on the same stream as without the
// pipeline: results are bit-identical.  One pass (or host_pipeline = 0): everything on the compute stream, as before.
template <class Up, class Run, class Down>
int host_passes(reid_ctx* ctx, int n, int pass, Up up, Run run, Down down) {
    const int passes = (n + pass - 1) / pass;
    if (passes <= 1 || !ctx->host_pipeline) {
        REID_TRY(up(0, n, ctx->stream));
        for (int i = 0; i < n; i += pass) REID_TRY(run(i, n - i < pass ? n - i : pass));
        REID_TRY(down(0, n, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return REID_OK;
    }
    REID_TRY(ctx_pipe_events(ctx, passes));
    hipStream_t cs = ctx->copy_stream;
    hipEvent_t* up_ev = ctx->pipe_ev.data();
    hipEvent_t* run_ev = ctx->pipe_ev.data() + passes;
    auto span = [&](int k, int* i, int* m) { *i = k * pass; *m = n - *i < pass ? n - *i : pass; };
    int i, m, rc = REID_OK;
    span(0, &i, &m);
    rc = up(i, m, cs);
    if (rc == REID_OK && hipEventRecord(up_ev[0], cs) != hipSuccess) rc = REID_ERR_HIP;
    for (int k = 0; k < passes && rc == REID_OK; ++k) {
        span(k, &i, &m);
        if (hipStreamWaitEvent(ctx->stream, up_ev[k], 0) != hipSuccess) { rc = REID_ERR_HIP; break; }
        if ((rc = run(i, m)) != REID_OK) break;
        if (hipEventRecord(run_ev[k], ctx->stream) != hipSuccess) { rc = REID_ERR_HIP; break; }
        if (k + 1 < passes) {
            span(k + 1, &i, &m);
            if ((rc = up(i, m, cs)) != REID_OK) break;
            if (hipEventRecord(up_ev[k + 1], cs) != hipSuccess) { rc = REID_ERR_HIP; break; }
        }
        if (k >= 1) {
            span(k - 1, &i, &m);
            if (hipStreamWaitEvent(cs, run_ev[k - 1], 0) != hipSuccess) { rc = REID_ERR_HIP; break; }
            if ((rc = down(i, m, cs)) != REID_OK) break;
        }
    }
    if (rc == REID_OK) {
        span(passes - 1, &i, &m);
        if (hipStreamWaitEvent(cs, run_ev[passes - 1], 0) != hipSuccess) rc = REID_ERR_HIP;
        else rc = down(i, m, cs);
    }
    // both streams are drained on every path: the caller's buffers must not be touched after the call returns
    const hipError_t e1 = hipStreamSynchronize(cs), e2 = hipStreamSynchronize(ctx->stream);
    if (rc == REID_OK && (e1 != hipSuccess || e2 != hipSuccess)) {
        reid_set_error("host_passes: stream synchronisation -> %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        rc = REID_ERR_HIP;
    }
    return rc;
}
int embed_ragged_enqueue(reid_ctx* ctx, const char* tag, const uint8_t* packed, const int64_t* offsets, const int32_t* hw, int n,
                         float** d_emb_out, float** d_log_out, bool side_copy);   // api.hip: upload + resize + forward, no synchronisation
void prof_begin(reid_ctx* ctx, int kind, double flops, double bytes);
void prof_end(reid_ctx* ctx);

#define LAUNCH_CHECK()                                                                 \
    do {                                                                               \
        hipError_t _e = hipGetLastError();                                             \
        if (_e != hipSuccess) {                                                        \
            reid_set_error("%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return REID_ERR_HIP;                                                       \
        }                                                                              \
    } while (0)
