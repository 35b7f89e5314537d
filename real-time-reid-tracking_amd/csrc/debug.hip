// Kernel experiments and correctness harnesses (include/reid_hip_debug.h).  Built into libreid_hip_debug.so, which links
// against libreid_hip.so: nothing here is part of the product library or of the drop-in C ABI.
#include "reid_internal.h"
#include <string.h>
#include <stdlib.h>
#include <vector>

// ------------------------------------------------------------------------------------------------ experiment switches
// Every kernel-selection / arithmetic-form / summation-order switch of a context, by the name of its reid_ctx field.  Until round 5
// these were environment variables read by reid_ctx_create in the PRODUCT library (a stray REID_* in a tracker's environment
// silently changed embeddings); now the product library has fixed defaults and only this library can move them.
namespace {
struct Switch { const char* name; int reid_ctx::* field; };
const Switch kSwitches[] = {
    {"f16_cfg", &reid_ctx::f16_cfg}, {"f16_lin_256", &reid_ctx::f16_lin_256}, {"f16_split_k", &reid_ctx::f16_split_k},
    {"bank_fast", &reid_ctx::bank_fast}, {"side_copy", &reid_ctx::side_copy}, {"f32_stem_pool", &reid_ctx::f32_stem_pool},
    {"stem_split", &reid_ctx::stem_split}, {"split_pair", &reid_ctx::split_pair}, {"split_lean_epi", &reid_ctx::split_lean_epi},
    {"knn_wide", &reid_ctx::knn_wide}, {"f16_loader_prio", &reid_ctx::f16_loader_prio}, {"f16_frag_ahead", &reid_ctx::f16_frag_ahead},
    {"pack_epilogue", &reid_ctx::pack_epilogue}, {"f16_wide_splitk", &reid_ctx::f16_wide_splitk}, {"f32_split_k", &reid_ctx::f32_split_k},
    {"swin_fold", &reid_ctx::swin_fold}, {"swin_stop", &reid_ctx::swin_stop}, {"select_two_pass", &reid_ctx::select_two_pass},
    {"split_terms", &reid_ctx::split_terms}, {"f32_conv", &reid_ctx::f32_conv}, {"swin_attn_mfma", &reid_ctx::swin_attn_mfma},
    {"swin_attn_split", &reid_ctx::swin_attn_split}, {"swin_two_linear", &reid_ctx::swin_two_linear},
    {"f16_loader_waves", &reid_ctx::f16_loader_waves}, {"f16_halo", &reid_ctx::f16_halo}, {"f16_stem_fused", &reid_ctx::f16_stem_fused},
    {"f16_se_tail", &reid_ctx::f16_se_tail}, {"f16_c64", &reid_ctx::f16_c64}, {"swin_chunk_cap", &reid_ctx::swin_chunk_cap},
    {"split_x3", &reid_ctx::split_x3}, {"x3_ablate", &reid_ctx::x3_ablate}, {"x3_unroll", &reid_ctx::x3_unroll}, {"x3_narrow", &reid_ctx::x3_narrow}, {"conv_x3s", &reid_ctx::conv_x3s}, {"x3s_sk_cap", &reid_ctx::x3s_sk_cap}, {"x3_l4_narrow_nmt", &reid_ctx::x3_l4_narrow_nmt}, {"split_x3_small", &reid_ctx::split_x3_small}, {"split_x3_min_blocks", &reid_ctx::split_x3_min_blocks}, {"f32_dist_bk16", &reid_ctx::f32_dist_bk16}, {"lin_x3", &reid_ctx::lin_x3},
    {"host_pipeline", &reid_ctx::host_pipeline}, {"x3_sk_cap", &reid_ctx::x3_sk_cap}, {"chain", &reid_ctx::chain}, {"split_gemm_min_tiles", &reid_ctx::split_gemm_min_tiles},
};
}  // namespace

extern "C" int reid_debug_set_switch(reid_ctx* ctx, const char* name, long long value) {
    ARG_CHECK(ctx && name);
    CTX_GUARD(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));       // nothing in flight may see the switch move
    if (!strcmp(name, "knn_wide_min")) {
        ctx->knn_wide_min = value;
        return REID_OK;
    }
    for (const Switch& s : kSwitches)
        if (!strcmp(name, s.name)) {
            if (!strcmp(name, "split_terms") && value != 3 && value != 4) break;
            ctx->*(s.field) = (int)value;
            return REID_OK;
        }
    reid_set_error("reid_debug_set_switch: no switch '%s' (or a value it does not take)", name);
    return REID_ERR_ARG;
}

extern "C" int reid_debug_get_switch(reid_ctx* ctx, const char* name, long long* value) {
    ARG_CHECK(ctx && name && value);
    if (!strcmp(name, "knn_wide_min")) {
        *value = ctx->knn_wide_min;
        return REID_OK;
    }
    for (const Switch& s : kSwitches)
        if (!strcmp(name, s.name)) {
            *value = ctx->*(s.field);
            return REID_OK;
        }
    reid_set_error("reid_debug_get_switch: no switch '%s'", name);
    return REID_ERR_ARG;
}

// ------------------------------------------------------------------------------------------------ kernel experiments
// Times `iters` launches of one fp16 implicit-GEMM convolution on random device data (not part of the public header).
extern "C" int reid_debug_conv_f16(reid_ctx* ctx, int n, int h, int w, int cin, int cout, int r, int stride, int pad, int cfg,
                                   int iters, float* ms_per_launch) {
    ARG_CHECK(ctx && ms_per_launch && ctx->se18.loaded);
    CTX_GUARD(ctx);
    typedef _Float16 f16;
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (w + 2 * pad - r) / stride + 1;
    const size_t nin = (size_t)n * h * w * cin, nw = (size_t)cout * r * r * cin, nout = (size_t)n * ho * wo * cout;
    f16 *x, *wt, *out;
    REID_TRY(ctx_ws(ctx, "dbg.x", nin * 2, (void**)&x));
    REID_TRY(ctx_ws(ctx, "dbg.w", nw * 2, (void**)&wt));
    REID_TRY(ctx_ws(ctx, "dbg.out", nout * 2, (void**)&out));
    // random-ish operands: the loaded weight blob (f32 -> f16), cycled
    const size_t src_n = ctx->se18.n_floats;
    for (size_t o = 0; o < nin; o += src_n) REID_TRY(launch_f32_to_f16(ctx, ctx->se18.blob, nin - o < src_n ? nin - o : src_n, x + o));
    for (size_t o = 0; o < nw; o += src_n) REID_TRY(launch_f32_to_f16(ctx, ctx->se18.blob, nw - o < src_n ? nw - o : src_n, wt + o));
    if (getenv("REID_DEBUG_ZERO")) {   // clock experiment: all-zero operands draw less power (DVFS give-back)
        HIP_TRY(hipMemsetAsync(x, 0, nin * 2, ctx->stream));
        HIP_TRY(hipMemsetAsync(wt, 0, nw * 2, ctx->stream));
    }
    const int c0 = ctx->f16_cfg;
    const int h0 = ctx->f16_halo;
    ctx->f16_halo = cfg >= 2000000 ? 2 : 0;   // 2xxxxxx: force the LDS-halo kernel, 2000001: with loader waves
    const int l0 = ctx->f16_loader_waves;
    if (cfg >= 2000000) ctx->f16_loader_waves = cfg & 1;
    ctx->f16_cfg = cfg >= 2000000 ? 0 : cfg;
    int st = REID_OK;
    for (int i = 0; i < 2 && st == REID_OK; ++i)
        st = conv_gemm16(ctx, A16_IM2COL, x, n, h, w, cin, wt, cout, r, r, stride, pad, r * r * cin, nullptr, nullptr, nullptr, 0, nullptr, out);
    if (st == REID_OK) st = reid_timer_start(ctx);
    for (int i = 0; i < iters && st == REID_OK; ++i)
        st = conv_gemm16(ctx, A16_IM2COL, x, n, h, w, cin, wt, cout, r, r, stride, pad, r * r * cin, nullptr, nullptr, nullptr, 0, nullptr, out);
    float ms = 0.f;
    if (st == REID_OK) st = reid_timer_stop(ctx, &ms);
    ctx->f16_cfg = c0;
    ctx->f16_halo = h0;
    ctx->f16_loader_waves = l0;
    *ms_per_launch = ms / (iters > 0 ? iters : 1);
    return st;
}


// Times `iters` launches of one fp32-class 3x3 stride-1 convolution (SPLIT build of the LDS-halo kernel: c real channels as
// [xh | xl'] against [wh 2^11 | wh | wl'] weights, fp32 out with BN + ReLU epilogue) on operands cut from the loaded weight blob.
// ablate: experiment switches of the PAIR loop (Gemm16Params.ablate; results are then wrong): 1 no weight DMA after the first step,
// 2 no halo DMA after the first phase, 4 no MFMAs, 8 no fragment reads, 16 no block barriers.
extern "C" int reid_debug_conv_split(reid_ctx* ctx, int n, int h, int w, int c, int cout, int ablate, int iters, float* ms_per_launch) {
    ARG_CHECK(ctx && ms_per_launch && ctx->se18.loaded && c % 64 == 0 && cout % 64 == 0);
    CTX_GUARD(ctx);
    typedef _Float16 f16;
    const size_t nin = (size_t)n * h * w * 2 * c, nw = (size_t)cout * 9 * 3 * c, nout = (size_t)n * h * w * cout;
    f16 *x, *wt;
    float *out, *sc;
    REID_TRY(ctx_ws(ctx, "dbgs.x", nin * 2, (void**)&x));
    REID_TRY(ctx_ws(ctx, "dbgs.w", nw * 2, (void**)&wt));
    REID_TRY(ctx_ws(ctx, "dbgs.out", nout * 4, (void**)&out));
    REID_TRY(ctx_ws(ctx, "dbgs.sc", (size_t)cout * 2 * 4, (void**)&sc));
    const size_t src_n = ctx->se18.n_floats;
    for (size_t o = 0; o < nin; o += src_n) REID_TRY(launch_f32_to_f16(ctx, ctx->se18.blob, nin - o < src_n ? nin - o : src_n, x + o));
    for (size_t o = 0; o < nw; o += src_n) REID_TRY(launch_f32_to_f16(ctx, ctx->se18.blob, nw - o < src_n ? nw - o : src_n, wt + o));
    HIP_TRY(hipMemcpyAsync(sc, ctx->se18.blob, (size_t)cout * 2 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (ablate & 128) {   // clock experiment: all-zero operands draw less power (is the kernel power-limited?)
        HIP_TRY(hipMemsetAsync(x, 0, nin * 2, ctx->stream));
        HIP_TRY(hipMemsetAsync(wt, 0, nw * 2, ctx->stream));
    }
    Gemm16Params q;
    memset(&q, 0, sizeof(q));
    q.split_terms = 3;
    q.H = h; q.W = w; q.Cin = 3 * c; q.R = 3; q.S = 3; q.stride = 1; q.pad = 1; q.Ho = h; q.Wo = w;
    q.M = n * h * w; q.N = cout; q.K = 27 * c; q.ldb = q.K;
    q.A = x; q.B = wt; q.C32 = out; q.ldc = cout;
    q.col_scale = sc; q.col_shift = sc + cout; q.relu = 1;
    q.acc_scale = 1.0f / 2048.0f;
    q.zero_page = ctx->se18.zero_page;
    q.ablate = ablate & 127;
    q.diag = ctx->conv_diag;
    int st = REID_OK;
    for (int i = 0; i < 2 && st == REID_OK; ++i) st = launch_conv3x3_split(ctx, q, REID_K_CONV_GEMM, 0, 0);
    if (st == REID_OK) st = reid_timer_start(ctx);
    for (int i = 0; i < iters && st == REID_OK; ++i) st = launch_conv3x3_split(ctx, q, REID_K_CONV_GEMM, 0, 0);
    float ms = 0.f;
    if (st == REID_OK) st = reid_timer_stop(ctx, &ms);
    *ms_per_launch = ms / (iters > 0 ? iters : 1);
    return st;
}

// Correctness harness for conv3x3_c64_f16.hip (tests only, not part of the C ABI): fp32 host operands are rounded to f16,
// the kernel runs once, the f16 result and the fp32 per-image statistics come back as fp32.
extern "C" int reid_debug_conv_c64(reid_ctx* ctx, int n, const float* x, const float* w_krsc, const float* scale,
                                   const float* shift, const float* residual, int relu, float* out, float* stats) {
    ARG_CHECK(ctx && x && w_krsc && out && n >= 1);
    CTX_GUARD(ctx);
    typedef _Float16 f16;
    const size_t nact = (size_t)n * 64 * 32 * 64, nw = 64 * 576;
    float *xf, *wf, *rf = nullptr, *sc = nullptr, *sh = nullptr, *st = nullptr, *of;
    f16 *xh, *wh, *rh = nullptr, *oh, *zp;
    REID_TRY(ctx_ws(ctx, "dbg64.xf", nact * 4, (void**)&xf));
    REID_TRY(ctx_ws(ctx, "dbg64.wf", nw * 4, (void**)&wf));
    REID_TRY(ctx_ws(ctx, "dbg64.xh", nact * 2, (void**)&xh));
    REID_TRY(ctx_ws(ctx, "dbg64.wh", nw * 2, (void**)&wh));
    REID_TRY(ctx_ws(ctx, "dbg64.oh", nact * 2, (void**)&oh));
    REID_TRY(ctx_ws(ctx, "dbg64.of", nact * 4, (void**)&of));
    REID_TRY(ctx_ws(ctx, "dbg64.zp", 256, (void**)&zp));
    REID_TRY(ctx_ws(ctx, "dbg64.sc", 64 * 4, (void**)&sc));
    REID_TRY(ctx_ws(ctx, "dbg64.sh", 64 * 4, (void**)&sh));
    REID_TRY(ctx_ws(ctx, "dbg64.st", (size_t)n * 128 * 4, (void**)&st));
    HIP_TRY(hipMemsetAsync(zp, 0, 256, ctx->stream));
    HIP_TRY(hipMemcpyAsync(xf, x, nact * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(wf, w_krsc, nw * 4, hipMemcpyHostToDevice, ctx->stream));
    std::vector<float> ones(64, 1.f);
    HIP_TRY(hipMemcpyAsync(sc, scale ? scale : ones.data(), 64 * 4, hipMemcpyHostToDevice, ctx->stream));
    if (shift) HIP_TRY(hipMemcpyAsync(sh, shift, 64 * 4, hipMemcpyHostToDevice, ctx->stream));
    REID_TRY(launch_f32_to_f16(ctx, xf, nact, xh));
    REID_TRY(launch_scale_rows_f16(ctx, wf, sc, 64, 576, wh));
    if (residual) {
        REID_TRY(ctx_ws(ctx, "dbg64.rf", nact * 4, (void**)&rf));
        REID_TRY(ctx_ws(ctx, "dbg64.rh", nact * 2, (void**)&rh));
        HIP_TRY(hipMemcpyAsync(rf, residual, nact * 4, hipMemcpyHostToDevice, ctx->stream));
        REID_TRY(launch_f32_to_f16(ctx, rf, nact, rh));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // `ones` is a local
    REID_TRY(launch_conv3x3_c64_f16(ctx, xh, n, wh, shift ? sh : nullptr, rh, relu, stats ? st : nullptr, oh, zp));
    std::vector<f16> tmp(nact);
    HIP_TRY(hipMemcpyAsync(tmp.data(), oh, nact * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (stats) HIP_TRY(hipMemcpyAsync(stats, st, (size_t)n * 128 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < nact; ++i) out[i] = (float)tmp[i];
    return REID_OK;
}

// Times one dense fp16 GEMM C[m][n] = A[m][k] . B[n][k]^T (experiments: separates the im2col gather from the tile loop).
extern "C" int reid_debug_gemm_f16(reid_ctx* ctx, int m, int n, int k, int cfg, int iters, float* ms_per_launch,
                                   unsigned long long* diag_host /* [64*8*4] or NULL */) {
    ARG_CHECK(ctx && ms_per_launch && ctx->se18.loaded);
    CTX_GUARD(ctx);
    unsigned long long* d_diag = nullptr;
    if (diag_host) {
        REID_TRY(ctx_ws(ctx, "dbg.diag", 64 * 8 * 4 * 8, (void**)&d_diag));
        HIP_TRY(hipMemsetAsync(d_diag, 0, 64 * 8 * 4 * 8, ctx->stream));
    }
    typedef _Float16 f16;
    f16 *a, *b, *c;
    REID_TRY(ctx_ws(ctx, "dbg.x", (size_t)m * k * 2, (void**)&a));
    REID_TRY(ctx_ws(ctx, "dbg.w", (size_t)n * k * 2, (void**)&b));
    REID_TRY(ctx_ws(ctx, "dbg.out", (size_t)m * n * 2, (void**)&c));
    const size_t src_n = ctx->se18.n_floats;
    for (size_t o = 0; o < (size_t)m * k; o += src_n)
        REID_TRY(launch_f32_to_f16(ctx, ctx->se18.blob, (size_t)m * k - o < src_n ? (size_t)m * k - o : src_n, a + o));
    for (size_t o = 0; o < (size_t)n * k; o += src_n)
        REID_TRY(launch_f32_to_f16(ctx, ctx->se18.blob, (size_t)n * k - o < src_n ? (size_t)n * k - o : src_n, b + o));
    Gemm16Params p;
    memset(&p, 0, sizeof(p));
    p.A = a; p.lda = k; p.B = b; p.ldb = k; p.M = m; p.N = n; p.K = k; p.C = c; p.ldc = n;
    p.zero_page = ctx->se18.zero_page;
    p.diag = d_diag;
    const int c0 = ctx->f16_cfg;
    ctx->f16_cfg = cfg;
    int st = REID_OK;
    for (int i = 0; i < 2 && st == REID_OK; ++i) st = launch_gemm_f16(ctx, A16_DENSE, p, REID_K_CONV_GEMM, 0, 0);
    if (st == REID_OK) st = reid_timer_start(ctx);
    for (int i = 0; i < iters && st == REID_OK; ++i) st = launch_gemm_f16(ctx, A16_DENSE, p, REID_K_CONV_GEMM, 0, 0);
    float ms = 0.f;
    if (st == REID_OK) st = reid_timer_stop(ctx, &ms);
    ctx->f16_cfg = c0;
    *ms_per_launch = ms / (iters > 0 ? iters : 1);
    if (st == REID_OK && diag_host) {
        HIP_TRY(hipMemcpyAsync(diag_host, d_diag, 64 * 8 * 4 * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return st;
}

extern "C" int reid_debug_conv_diag(reid_ctx* ctx, int enable, unsigned long long* out_host /* [64*8*4] when disabling */) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    if (enable) {
        REID_TRY(ctx_ws(ctx, "dbg.cdiag", 64 * 8 * 5 * 8, (void**)&ctx->conv_diag));
        HIP_TRY(hipMemsetAsync(ctx->conv_diag, 0, 64 * 8 * 5 * 8, ctx->stream));
    } else {
        if (out_host && ctx->conv_diag) {
            HIP_TRY(hipMemcpyAsync(out_host, ctx->conv_diag, 64 * 8 * 5 * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        ctx->conv_diag = nullptr;
    }
    return REID_OK;
}

// Times `iters` launches of one exact-fp32 convolution of the ResNet18-SE path on random device data.
// flags: 1 = fused input affine + ReLU (conv2 of an SE block), 2 = BN scale/shift epilogue, 4 = residual + ReLU, 8 = statistics.
// variant: 0 = gemm_f32_kernel<A_IM2COL> (round-1 kernel), 1 = conv_f32.hip.
extern "C" int reid_debug_conv_f32(reid_ctx* ctx, int n, int h, int w, int cin, int cout, int r, int stride, int pad, int flags,
                                   int variant, int iters, float* ms_per_launch) {
    ARG_CHECK(ctx && ms_per_launch && ctx->se18.loaded && n >= 1);
    CTX_GUARD(ctx);
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (w + 2 * pad - r) / stride + 1;
    const size_t nin = (size_t)n * h * w * cin, nw = (size_t)cout * r * r * cin, nout = (size_t)n * ho * wo * cout;
    float *x, *wt, *out, *res, *asc, *ash, *stats;
    REID_TRY(ctx_ws(ctx, "dbg32.x", nin * 4, (void**)&x));
    REID_TRY(ctx_ws(ctx, "dbg32.w", nw * 4, (void**)&wt));
    REID_TRY(ctx_ws(ctx, "dbg32.out", nout * 4, (void**)&out));
    REID_TRY(ctx_ws(ctx, "dbg32.res", nout * 4, (void**)&res));
    REID_TRY(ctx_ws(ctx, "dbg32.asc", (size_t)n * cin * 4, (void**)&asc));
    REID_TRY(ctx_ws(ctx, "dbg32.ash", (size_t)n * cin * 4, (void**)&ash));
    REID_TRY(ctx_ws(ctx, "dbg32.stats", (size_t)n * 2048 * 4 * 16, (void**)&stats));
    // random-ish operands: the loaded weight blob, cycled
    const size_t src_n = ctx->se18.n_floats;
    auto fill = [&](float* dst, size_t cnt) -> int {
        for (size_t o = 0; o < cnt; o += src_n)
            HIP_TRY(hipMemcpyAsync(dst + o, ctx->se18.blob, (cnt - o < src_n ? cnt - o : src_n) * 4, hipMemcpyDeviceToDevice, ctx->stream));
        return REID_OK;
    };
    REID_TRY(fill(x, nin));
    REID_TRY(fill(wt, nw));
    REID_TRY(fill(res, nout));
    REID_TRY(fill(asc, (size_t)n * cin));
    REID_TRY(fill(ash, (size_t)n * cin));
    const int v0 = ctx->f32_conv;
    ctx->f32_conv = variant;
    const float* cs = (flags & 2) ? ctx->se18.neck_scale : nullptr;   // any 512 floats
    const float* sh = (flags & 2) ? ctx->se18.neck_shift : nullptr;
    auto run = [&]() {
        return conv_gemm(ctx, A_IM2COL, x, n, h, w, cin, wt, cout, r, r, stride, pad, r * r * cin, (flags & 1) ? asc : nullptr,
                         (flags & 1) ? ash : nullptr, (flags & 1), cs, sh, (flags & 4) ? res : nullptr, (flags & 4) ? 1 : 0,
                         (flags & 8) ? stats : nullptr, out);
    };
    int st = REID_OK;
    for (int i = 0; i < 2 && st == REID_OK; ++i) st = run();
    if (st == REID_OK) st = reid_timer_start(ctx);
    for (int i = 0; i < iters && st == REID_OK; ++i) st = run();
    float ms = 0.f;
    if (st == REID_OK) st = reid_timer_stop(ctx, &ms);
    ctx->f32_conv = v0;
    *ms_per_launch = ms / (iters > 0 ? iters : 1);
    return st;
}

// The device k-way merge of reid_knn_gallery_sharded_dev on host lists (tests: any number of virtual shards on one GPU).
extern "C" int reid_debug_knn_merge(reid_ctx* ctx, const float* Dall, const int32_t* Iall, int world, int nq, int kk, int k, float* D,
                                    int32_t* I) {
    ARG_CHECK(ctx && Dall && Iall && D && I && world >= 1 && nq >= 1 && kk >= 1 && k >= 1);
    CTX_GUARD(ctx);
    float *dD, *oD;
    int32_t *dI, *oI;
    const size_t cnt = (size_t)world * nq * kk;
    REID_TRY(ctx_ws(ctx, "dbg.mD", cnt * 4, (void**)&dD));
    REID_TRY(ctx_ws(ctx, "dbg.mI", cnt * 4, (void**)&dI));
    REID_TRY(ctx_ws(ctx, "dbg.moD", (size_t)nq * k * 4, (void**)&oD));
    REID_TRY(ctx_ws(ctx, "dbg.moI", (size_t)nq * k * 4, (void**)&oI));
    HIP_TRY(hipMemcpyAsync(dD, Dall, cnt * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dI, Iall, cnt * 4, hipMemcpyHostToDevice, ctx->stream));
    REID_TRY(launch_knn_merge(ctx, dD, dI, world, nq, kk, k, oD, oI));
    HIP_TRY(hipMemcpyAsync(D, oD, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(I, oI, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}

// One Swin Linear layer (swin_transformer.py:23-39, 191-232) on random device data: mode bit 0 = fp16-storage GEMM (else exact
// fp32), bits 1-2 = epilogue: 0 bias, 1 bias + erf-GELU, 2 bias + fp32 residual into the fp32 stream.
extern "C" int reid_debug_linear(reid_ctx* ctx, int m, int n, int k, int mode, int iters, float* ms_per_launch) {
    ARG_CHECK(ctx && ms_per_launch && ctx->se18.loaded && m > 0 && n > 0 && k > 0);
    CTX_GUARD(ctx);
    typedef _Float16 f16;
    const bool h = mode & 1;
    const int epi = (mode >> 1) & 3;
    const int npad = (n + 63) / 64 * 64;
    float *a32, *b32, *bias, *res, *c32;
    f16 *a16, *b16, *c16;
    REID_TRY(ctx_ws(ctx, "dbg.x", (size_t)m * k * 4, (void**)&a32));
    REID_TRY(ctx_ws(ctx, "dbg.w", (size_t)npad * k * 4, (void**)&b32));
    REID_TRY(ctx_ws(ctx, "dbg.bias", (size_t)npad * 4, (void**)&bias));
    REID_TRY(ctx_ws(ctx, "dbg.res", (size_t)m * n * 4, (void**)&res));
    REID_TRY(ctx_ws(ctx, "dbg.out", (size_t)m * n * 4, (void**)&c32));
    REID_TRY(ctx_ws(ctx, "dbg.x16", (size_t)m * k * 2, (void**)&a16));
    REID_TRY(ctx_ws(ctx, "dbg.w16", (size_t)npad * k * 2, (void**)&b16));
    REID_TRY(ctx_ws(ctx, "dbg.out16", (size_t)m * n * 2, (void**)&c16));
    const size_t src_n = ctx->se18.n_floats;
    for (size_t o = 0; o < (size_t)m * k; o += src_n) {
        const size_t cnt = (size_t)m * k - o < src_n ? (size_t)m * k - o : src_n;
        HIP_TRY(hipMemcpyAsync(a32 + o, ctx->se18.blob, cnt * 4, hipMemcpyDeviceToDevice, ctx->stream));
        REID_TRY(launch_f32_to_f16(ctx, ctx->se18.blob, cnt, a16 + o));
    }
    for (size_t o = 0; o < (size_t)npad * k; o += src_n) {
        const size_t cnt = (size_t)npad * k - o < src_n ? (size_t)npad * k - o : src_n;
        HIP_TRY(hipMemcpyAsync(b32 + o, ctx->se18.blob, cnt * 4, hipMemcpyDeviceToDevice, ctx->stream));
        REID_TRY(launch_f32_to_f16(ctx, ctx->se18.blob, cnt, b16 + o));
    }
    HIP_TRY(hipMemsetAsync(bias, 0, (size_t)npad * 4, ctx->stream));
    HIP_TRY(hipMemsetAsync(res, 0, (size_t)m * n * 4, ctx->stream));
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = a32; p.lda = k; p.B = b32; p.ldb = k; p.M = m; p.N = n; p.K = k; p.C = c32; p.ldc = n;
    p.col_shift = bias; p.act = epi == 1; p.residual = epi == 2 ? res : nullptr;
    Gemm16Params q;
    memset(&q, 0, sizeof(q));
    q.A = a16; q.lda = k; q.B = b16; q.ldb = k; q.M = m; q.N = npad; q.K = k; q.ldc = n;
    q.C = epi == 2 ? nullptr : c16; q.C32 = epi == 2 ? c32 : nullptr; q.res32 = epi == 2 ? res : nullptr;
    q.col_shift = bias; q.lin = 1; q.act = epi == 1; q.n_real = n;
    q.zero_page = ctx->se18.zero_page;
    auto run = [&]() { return h ? launch_gemm_f16(ctx, A16_DENSE, q, REID_K_CONV_GEMM, 0, 0) : launch_gemm_f32(ctx, A_DENSE, E_BIAS, p, REID_K_CONV_GEMM, 0, 0); };
    int st = REID_OK;
    for (int i = 0; i < 2 && st == REID_OK; ++i) st = run();
    if (st == REID_OK) st = reid_timer_start(ctx);
    for (int i = 0; i < iters && st == REID_OK; ++i) st = run();
    float ms = 0.f;
    if (st == REID_OK) st = reid_timer_stop(ctx, &ms);
    *ms_per_launch = ms / (iters > 0 ? iters : 1);
    return st;
}

// Experiment switch of the fused distance + selection kernel (dist_select.hip: 1 = no filter phase, 2 = no list writes - both
// leave results incomplete -, 4 = print candidate-list statistics).  Lives here so that no environment variable can thin out the
// product's k-NN.
extern "C" int reid_debug_select_exp(reid_ctx* ctx, int mode) {
    ARG_CHECK(ctx && mode >= 0 && mode <= 4);
    ctx->select_exp = mode;
    return REID_OK;
}

// Test switch of the large k-NN path (knn_wide.hip): enable = 0 sends every search through the fused fp32 kernel; force > 0 makes
// every row whose index is a multiple of it take the exact-row fallback.
extern "C" int reid_debug_knn_wide(reid_ctx* ctx, int enable, int force) {
    ARG_CHECK(ctx && force >= 0);
    ctx->knn_wide = enable != 0;
    ctx->knn_wide_force = force;
    return REID_OK;
}

// One Swin Linear layer on HOST operands through the f16 linear build of gemm_f16.hip, results back on the host (correctness
// harness: row-position invariance, tests/test_gpu_parity.py).  mode 1 = fp16 storage (operands rounded to f16), 2 = fp32-class
// (x packed to [xh | xl'], weights to [wh 2^11 | wh | wl'], K = 3 k virtual columns).  flags bit 0 = erf-GELU; bit 1 = f16 output
// through the LDS-staged epilogue (mode 1: plain f16; mode 2: [yh | yl'], returned as yh + yl' / 2^11), else fp32 output through
// the buffer-store epilogue with `res` (may be null) added.  out: [m][n] fp32.
extern "C" int reid_debug_linear_rows(reid_ctx* ctx, const float* x, const float* w, const float* bias, const float* res, int m, int n,
                                      int k, int mode, int flags, float* out) {
    ARG_CHECK(ctx && x && w && out && m > 0 && n > 0 && k > 0 && k % 32 == 0 && (mode == 1 || mode == 2) && ctx->se18.zero_page);
    CTX_GUARD(ctx);
    typedef _Float16 f16;
    const bool act = flags & 1, f16out = flags & 2;
    const int npad = (n + 63) / 64 * 64;
    float *x32, *w32, *b32, *r32, *c32;
    f16 *a16, *w16, *c16;
    REID_TRY(ctx_ws(ctx, "dbgr.x", (size_t)m * k * 4, (void**)&x32));
    REID_TRY(ctx_ws(ctx, "dbgr.w", (size_t)npad * k * 4, (void**)&w32));
    REID_TRY(ctx_ws(ctx, "dbgr.bias", (size_t)npad * 4, (void**)&b32));
    REID_TRY(ctx_ws(ctx, "dbgr.res", (size_t)m * n * 4, (void**)&r32));
    REID_TRY(ctx_ws(ctx, "dbgr.out", (size_t)m * n * 4, (void**)&c32));
    REID_TRY(ctx_ws(ctx, "dbgr.a16", (size_t)m * 2 * k * 2, (void**)&a16));
    REID_TRY(ctx_ws(ctx, "dbgr.w16", (size_t)npad * 3 * k * 2, (void**)&w16));
    REID_TRY(ctx_ws(ctx, "dbgr.c16", (size_t)m * 2 * n * 2, (void**)&c16));
    HIP_TRY(hipMemsetAsync(w32, 0, (size_t)npad * k * 4, ctx->stream));
    HIP_TRY(hipMemsetAsync(b32, 0, (size_t)npad * 4, ctx->stream));
    HIP_TRY(hipMemsetAsync(w16, 0, (size_t)npad * 3 * k * 2, ctx->stream));
    HIP_TRY(hipMemcpyAsync(x32, x, (size_t)m * k * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(w32, w, (size_t)n * k * 4, hipMemcpyHostToDevice, ctx->stream));
    if (bias) HIP_TRY(hipMemcpyAsync(b32, bias, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (res) HIP_TRY(hipMemcpyAsync(r32, res, (size_t)m * n * 4, hipMemcpyHostToDevice, ctx->stream));
    Gemm16Params q;
    memset(&q, 0, sizeof(q));
    q.M = m; q.N = npad; q.n_real = n; q.lin = 1; q.act = act; q.col_shift = b32; q.zero_page = ctx->se18.zero_page;
    if (mode == 1) {
        REID_TRY(launch_f32_to_f16(ctx, x32, (size_t)m * k, a16));
        REID_TRY(launch_f32_to_f16(ctx, w32, (size_t)npad * k, w16));
        q.A = a16; q.lda = k; q.B = w16; q.ldb = k; q.K = k;
    } else {
        REID_TRY(launch_split_pack(ctx, x32, m, k, a16));
        REID_TRY(launch_split_weights(ctx, w32, n, 1, k, 3, w16));
        q.A = a16; q.lda = 2 * k; q.B = w16; q.ldb = 3 * k; q.K = 3 * k;
        q.split_terms = 3; q.a_k = 2 * k; q.acc_scale = 1.0f / 2048.0f;
    }
    if (f16out) {
        q.C = c16;
        q.ldc = mode == 2 ? 2 * n : n;
        q.pack_out = mode == 2;
    } else {
        q.C32 = c32; q.ldc = n; q.res32 = res ? r32 : nullptr;
    }
    REID_TRY(launch_gemm_f16(ctx, A16_DENSE, q, REID_K_CONV_GEMM, 0, 0));
    if (!f16out) {
        HIP_TRY(hipMemcpyAsync(out, c32, (size_t)m * n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return REID_OK;
    }
    const size_t cols = mode == 2 ? 2 * (size_t)n : (size_t)n;
    std::vector<f16> h((size_t)m * cols);
    HIP_TRY(hipMemcpyAsync(h.data(), c16, h.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j)
            out[(size_t)i * n + j] = mode == 2 ? (float)h[i * cols + j] + (float)h[i * cols + n + j] * (1.0f / 2048.0f) : (float)h[i * cols + j];
    return REID_OK;
}

// Timing experiments on the fused pair of linears: bit 0 = no weight refills after the first two steps, bit 1 = no block barriers.
// The results are WRONG while a bit is set (which is why this lives here and not behind an environment variable of the library).
extern "C" int reid_debug_two_linear_ablate(reid_ctx* ctx, int bits) {
    ARG_CHECK(ctx && bits >= 0 && bits < 4);
    ctx->two_linear_ablate = bits;
    return REID_OK;
}

// The fused pair of linears (two_linear_f16.hip) on its own: out = res + w2 . act(w1 . x + b1) + b2 through launch_split_pack +
// launch_two_linear, x [m][c], w1 [hid][c], w2 [c][hid], res / out [m][c] fp32 on the host.  iters > 1 repeats the launch and
// returns the mean time in *ms (may be null).  The context must be in precision 2.
extern "C" int reid_debug_two_linear(reid_ctx* ctx, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                     const float* res, int m, int c, int hid, int act, int iters, float* out, float* ms, const float* ln_g,
                                     const float* ln_b) {
    ARG_CHECK(ctx && x && w1 && b1 && w2 && b2 && res && out && m > 0 && iters >= 1);
    CTX_GUARD(ctx);
    ARG_CHECK(two_linear_supported(ctx, m, c, hid));
    typedef _Float16 f16;
    float *x32, *dw1, *dw2, *db1, *db2, *r32, *o32;
    f16* a16;
    REID_TRY(ctx_ws(ctx, "dbg2.x", (size_t)m * c * 4, (void**)&x32));
    REID_TRY(ctx_ws(ctx, "dbg2.w1", (size_t)hid * c * 4, (void**)&dw1));
    REID_TRY(ctx_ws(ctx, "dbg2.w2", (size_t)hid * c * 4, (void**)&dw2));
    REID_TRY(ctx_ws(ctx, "dbg2.b1", (size_t)hid * 4, (void**)&db1));
    REID_TRY(ctx_ws(ctx, "dbg2.b2", (size_t)c * 4, (void**)&db2));
    REID_TRY(ctx_ws(ctx, "dbg2.res", (size_t)m * c * 4, (void**)&r32));
    REID_TRY(ctx_ws(ctx, "dbg2.out", (size_t)m * c * 4, (void**)&o32));
    REID_TRY(ctx_ws(ctx, "dbg2.a16", (size_t)m * 2 * c * 2, (void**)&a16));
    float *dg = nullptr, *dbt = nullptr;
    if (ln_g && ln_b) {   // LayerNorm(x) in the kernel instead of the packed input
        REID_TRY(ctx_ws(ctx, "dbg2.lng", (size_t)c * 4, (void**)&dg));
        REID_TRY(ctx_ws(ctx, "dbg2.lnb", (size_t)c * 4, (void**)&dbt));
        HIP_TRY(hipMemcpyAsync(dg, ln_g, (size_t)c * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(dbt, ln_b, (size_t)c * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    const _Float16* ain = dg ? nullptr : a16;
    HIP_TRY(hipMemcpyAsync(x32, x, (size_t)m * c * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dw1, w1, (size_t)hid * c * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dw2, w2, (size_t)hid * c * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db1, b1, (size_t)hid * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db2, b2, (size_t)c * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(r32, res, (size_t)m * c * 4, hipMemcpyHostToDevice, ctx->stream));
    REID_TRY(launch_split_pack(ctx, x32, m, c, a16));
    // the tiled weight images are cached by blob address: this harness re-uses its buffers, so drop what an earlier call left
    for (const void* key : {(const void*)((const char*)dw1 + 1), (const void*)((const char*)dw2 + 1)}) {
        auto it = ctx->split_w.find(key);
        if (it != ctx->split_w.end()) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            (void)hipFree(it->second);
            ctx->split_w.erase(it);
        }
    }
    REID_TRY(launch_two_linear(ctx, ain, m, c, hid, dw1, db1, dw2, db2, act, r32, o32, x32, dg, dbt));
    if (iters > 1) {
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, ctx->stream));
        for (int i = 0; i < iters; ++i) REID_TRY(launch_two_linear(ctx, ain, m, c, hid, dw1, db1, dw2, db2, act, r32, o32, x32, dg, dbt));
        HIP_TRY(hipEventRecord(e1, ctx->stream));
        HIP_TRY(hipEventSynchronize(e1));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        if (ms) *ms = t / iters;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    HIP_TRY(hipMemcpyAsync(out, o32, (size_t)m * c * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}

// ------------------------------------------------------------------------------------------------ loop-back communicator (tests)
// Several contexts of THIS process on ONE device act as the ranks of a job: every collective of csrc/comm.hip
// (reid_allgather_dev and what is built on it - ragged row gathers, reid_frame_gather, reid_knn_gallery_sharded_dev -
// reid_allreduce_f64) then runs with world > 1 on a one-GPU box, driven by one host thread per rank.  The exchange itself is
// a host rendezvous + device-to-device copies; RCCL is not involved (the product's only transport, reid_comm_init, is).
#include <condition_variable>
#include <chrono>
#include <mutex>

namespace {
struct LoopComm : reid_comm_loop {
    int world, attached;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long gen = 0;
    std::vector<const void*> send;
    std::vector<double> red;
    explicit LoopComm(int w) : world(w), attached(w), send(w, nullptr), red((size_t)w * 64, 0.0) {}

    // reusable barrier; gives up after 60 s (a rank that failed never arrives: the others must not hang the box)
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const long g = gen;
        if (++arrived == world) {
            arrived = 0;
            ++gen;
            cv.notify_all();
            return true;
        }
        return cv.wait_for(lk, std::chrono::seconds(60), [&] { return gen != g; });
    }
    int allgather(int rank, const void* d_send, void* d_recv, size_t bytes, hipStream_t st) override {
        HIP_TRY(hipStreamSynchronize(st));            // this rank's payload is complete
        send[rank] = d_send;
        if (!barrier()) { reid_set_error("loop-back all-gather: rank %d waited 60 s for the others", rank); return REID_ERR_STATE; }
        for (int r = 0; r < world; ++r)
            HIP_TRY(hipMemcpyAsync((char*)d_recv + (size_t)r * bytes, send[r], bytes, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (!barrier()) { reid_set_error("loop-back all-gather: rank %d waited 60 s for the others", rank); return REID_ERR_STATE; }
        return REID_OK;
    }
    int allreduce(int rank, double* inout, int count, int op) override {
        for (int i = 0; i < count; ++i) red[(size_t)rank * 64 + i] = inout[i];
        if (!barrier()) { reid_set_error("loop-back all-reduce: rank %d waited 60 s for the others", rank); return REID_ERR_STATE; }
        for (int i = 0; i < count; ++i) {
            double v = red[i];
            for (int r = 1; r < world; ++r) v = op == 0 ? v + red[(size_t)r * 64 + i] : (red[(size_t)r * 64 + i] > v ? red[(size_t)r * 64 + i] : v);
            inout[i] = v;
        }
        if (!barrier()) { reid_set_error("loop-back all-reduce: rank %d waited 60 s for the others", rank); return REID_ERR_STATE; }
        return REID_OK;
    }
    void detach(int) override {
        bool last;
        {
            std::lock_guard<std::mutex> lk(mu);
            last = --attached == 0;
        }
        if (last) delete this;
    }
};
}  // namespace

extern "C" int reid_debug_comm_loopback(reid_ctx** ctxs, int world) {
    ARG_CHECK(ctxs && world >= 1 && world <= 64);
    for (int r = 0; r < world; ++r) {
        ARG_CHECK(ctxs[r] && ctxs[r]->device == ctxs[0]->device);
        for (int q = 0; q < r; ++q) ARG_CHECK(ctxs[q] != ctxs[r]);
        if (ctxs[r]->comm) {
            reid_set_error("reid_debug_comm_loopback: context %d already has a communicator", r);
            return REID_ERR_STATE;
        }
    }
    LoopComm* lc = new LoopComm(world);
    for (int r = 0; r < world; ++r) {
        reid_comm* c = new reid_comm();
        c->rank = r;
        c->world = world;
        c->loop = lc;
        ctxs[r]->comm = c;
    }
    return REID_OK;
}
