/* Experiment and correctness-harness entry points of libreid_hip_debug.so (built from csrc/debug.hip + csrc/microbench.hip,
 * linked on top of libreid_hip.so).  NOT part of the drop-in C ABI of include/reid_hip.h: nothing the reference's callers
 * would bind lives here.  Used by the tools/ scripts (kernel A/B timing, feed / MFMA-shape microbenchmarks) and by one parity test
 * of the layer-1 fp16 convolution kernel.  All functions return a reid_hip.h status code. */
#pragma once
#include "reid_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Experiment switches of a context, by the name of the reid_ctx field (csrc/reid_internal.h: f16_cfg, split_pair, split_terms,
 * swin_two_linear, swin_attn_mfma, swin_attn_split, swin_fold, swin_stop, knn_wide, knn_wide_min, select_two_pass, f32_conv, ...).
 * They select kernels, arithmetic forms and summation orders; the product library gives them fixed defaults and reads none of them
 * from the environment.  The setter drains the context's stream first.  Unknown name / value: REID_ERR_ARG. */
int reid_debug_set_switch(reid_ctx* ctx, const char* name, long long value);
int reid_debug_get_switch(reid_ctx* ctx, const char* name, long long* value);
/* Times `iters` launches of one fp16-storage implicit-GEMM convolution (SERes18_IBN.py:120-128 shapes) on random device
 * data; cfg = BN*1000 + BK*10 + NST, 2000000 / 2000001 = LDS-halo kernel without / with loader waves. */
int reid_debug_conv_f16(reid_ctx* ctx, int n, int h, int w, int cin, int cout, int r, int stride, int pad, int cfg, int iters,
                        float* ms_per_launch);
/* Times one fp32-class 3x3 stride-1 convolution (SPLIT build of the LDS-halo kernel) on random device data; ablate != 0 switches
 * phases of its main loop off (1 weight DMA, 2 halo DMA, 4 MFMAs, 8 fragment reads, 16 barriers): timing experiments, wrong results. */
int reid_debug_conv_split(reid_ctx* ctx, int n, int h, int w, int c, int cout, int ablate, int iters, float* ms_per_launch);
/* The same for the exact-fp32 convolutions.  flags: 1 fused input affine + ReLU, 2 BN epilogue, 4 residual + ReLU,
 * 8 statistics; variant 0 = gemm_f32_kernel<A_IM2COL>, 1 = conv_f32.hip. */
int reid_debug_conv_f32(reid_ctx* ctx, int n, int h, int w, int cin, int cout, int r, int stride, int pad, int flags,
                        int variant, int iters, float* ms_per_launch);
/* Correctness harness of conv3x3_c64_f16.hip: fp32 host operands are rounded to f16, one launch, fp32 results back. */
int reid_debug_conv_c64(reid_ctx* ctx, int n, const float* x, const float* w_krsc, const float* scale, const float* shift,
                        const float* residual, int relu, float* out, float* stats);
/* Times one dense fp16 GEMM C[m][n] = A[m][k] . B[n][k]^T; diag_host: optional [64*8*4] per-wave cycle sums. */
int reid_debug_gemm_f16(reid_ctx* ctx, int m, int n, int k, int cfg, int iters, float* ms_per_launch,
                        unsigned long long* diag_host);
/* Times one Swin Linear layer [m][k] x [n][k]^T on random device data: mode bit 0 = fp16-storage GEMM (else exact fp32),
 * bits 1-2 = epilogue (0 bias, 1 bias + erf-GELU, 2 bias + fp32 residual into the fp32 stream). */
int reid_debug_linear(reid_ctx* ctx, int m, int n, int k, int mode, int iters, float* ms_per_launch);
/* One Swin Linear layer on host operands through the f16 linear build (correctness harness: identical input rows must give
 * bit-identical output rows wherever they sit in a tile).  mode 1 = fp16 storage, 2 = fp32-class (split operands); flags bit 0 =
 * erf-GELU, bit 1 = f16 output through the LDS-staged epilogue (mode 2: [yh | yl'], returned as yh + yl' / 2^11), else fp32
 * output (+ res) through the buffer-store epilogue. */
int reid_debug_linear_rows(reid_ctx* ctx, const float* x, const float* w, const float* bias, const float* res, int m, int n, int k,
                           int mode, int flags, float* out);
/* Timing experiments on that kernel (WRONG results while set): bit 0 = no weight refills after the first two steps, bit 1 = no block
 * barriers.  0 restores the product behaviour. */
int reid_debug_two_linear_ablate(reid_ctx* ctx, int bits);
/* The fused pair of linears of the fp32-class mode (csrc/two_linear_f16.hip) alone: out = res + w2 . act(w1 . x + b1) + b2, x / res /
 * out [m][c], w1 [hid][c], w2 [c][hid], all fp32 on the host; act 1 = erf-GELU.  iters > 1: the launch repeated, mean time in *ms.
 * ln_g / ln_b [c] (or both null): the first linear reads LayerNorm(x) (eps 1e-5), made in the kernel's prologue. */
int reid_debug_two_linear(reid_ctx* ctx, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                          const float* res, int m, int c, int hid, int act, int iters, float* out, float* ms, const float* ln_g,
                          const float* ln_b);
/* Experiment switch of the fused distance + selection kernel: 0 product behaviour, 1 / 2 skip phases (INCOMPLETE results: timing
 * only), 4 print candidate-list statistics. */
int reid_debug_select_exp(reid_ctx* ctx, int mode);
/* Test switch of the large k-NN path (candidates on the f16 matrix pipe + exact fp32 refinement): enable = 0 -> fused fp32 search
 * for every size; force > 0 -> rows whose index is a multiple of it take the exact-row fallback. */
int reid_debug_knn_wide(reid_ctx* ctx, int enable, int force);
/* Switches the s_memtime stamps of the loader-wave conv kernel on / off (out_host [64*8*5] when disabling). */
int reid_debug_conv_diag(reid_ctx* ctx, int enable, unsigned long long* out_host);
/* Bare MFMA loop with fragments re-read from LDS (shape 32 = 32x32x16 f16, 16 = 16x16x32 f16). */
int reid_debug_mfma_shape(reid_ctx* ctx, int shape, int iters, int blocks, float* tflops);
/* Registers-only MFMA loop (no LDS, no memory; 512 blocks x 4 waves): what the matrix pipe of THIS device sustains.  shape 32 =
 * v_mfma_f32_32x32x16_f16, 16 = v_mfma_f32_16x16x32_f16; zero != 0: all-zero operands (the chip holds its clock); else random. */
int reid_debug_mfma_bare(reid_ctx* ctx, int shape, int zero, int iters, float* tflops);
/* Operand-feed microbenchmark: rows of `rowb` bytes at `stride` from a `footprint`-byte buffer, LDS-DMA or register loads. */
int reid_debug_feed(reid_ctx* ctx, int mode, size_t footprint, int rowb, size_t stride, int iters, int inflight,
                    float* gbs_per_cu, float* tbs_chip);
/* The device k-way merge of reid_knn_gallery_sharded_dev run on host lists [world][nq][kk] (tests with virtual shards). */
int reid_debug_knn_merge(reid_ctx* ctx, const float* Dall, const int32_t* Iall, int world, int nq, int kk, int k, float* D,
                         int32_t* I);
/* Loop-back communicator: `world` contexts of this process on one device become ranks 0..world-1 of a job, one host thread
 * each; every collective of csrc/comm.hip then runs with world > 1 on a one-GPU box (host rendezvous + device copies, no
 * RCCL).  reid_comm_destroy / reid_ctx_destroy detach a rank. */
int reid_debug_comm_loopback(reid_ctx** ctxs, int world);
/* What a wave that stages data can issue beside the other wave's back-to-back v_mfma_f32_32x32x2_f32 (microbench.hip). */
int reid_debug_coissue(reid_ctx* ctx, int mode, int iters, int roles, double* cyc_mfma_wave, double* cyc_other_wave);

#ifdef __cplusplus
}
#endif
