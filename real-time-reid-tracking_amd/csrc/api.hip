// C ABI of libreid_hip.so (include/reid_hip.h): context, weights, the ResNet18-IBN-SE launch sequence,
// distance / selection entry points.  No torch types, no CPU compute fallback: every result comes from a HIP kernel.
#include "reid_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <sstream>

static thread_local char g_err[1024] = "";

void reid_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* reid_last_error(void) { return g_err; }

extern "C" int reid_device_count(int* n) {
    ARG_CHECK(n);
    HIP_TRY(hipGetDeviceCount(n));
    return REID_OK;
}

// ------------------------------------------------------------------------------------------------ context
extern "C" int reid_ctx_create(int device, reid_ctx** out) {
    ARG_CHECK(out);
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) {
        reid_set_error("reid_ctx_create: device %d not available (%d visible)", device, n);
        return REID_ERR_ARG;
    }
    DeviceGuard _dev_guard(device);   // streams / events below are created on `device`; the caller's current device is restored
    reid_ctx* c = new reid_ctx();
    c->device = device;
    // The product library reads exactly two environment variables, both SIZING knobs whose results are bit-identical by test
    // (tests/test_host_logic.py::test_product_library_reads_only_the_whitelisted_environment): the Swin pass cap and the pair count
    // from which k-NN takes the wide path.  Every switch that selects a kernel, an arithmetic form or a summation order is a field
    // of the context with a fixed default; experiments flip them through libreid_hip_debug.so (reid_debug_set_switch), never
    // through a tracker's environment.
    if (const char* e = getenv("REID_SWIN_CHUNK_MAX")) {
        const int v = atoi(e);
        if (v > 0 && v < c->swin_chunk_cap) c->swin_chunk_cap = v;
    }
    if (const char* e = getenv("REID_KNN_WIDE_MIN")) c->knn_wide_min = atoll(e);
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_TRY(hipHostMalloc((void**)&c->fault, 64, hipHostMallocMapped));   // the fault word kernels raise (reid_internal.h)
    memset(c->fault, 0, 64);
    HIP_TRY(hipEventCreate(&c->t0));
    HIP_TRY(hipEventCreate(&c->t1));
    *out = c;
    return REID_OK;
}

extern "C" int reid_ctx_destroy(reid_ctx* ctx) {
    if (!ctx) return REID_OK;
    DeviceGuard _dev_guard(ctx->device);
    hipStreamSynchronize(ctx->stream);
    comm_release(ctx);
    swin_release(ctx);
    for (int i = 0; i < 2; ++i)
        if (ctx->frame_ev[i]) hipEventDestroy(ctx->frame_ev[i]);
    if (ctx->copy_ev) hipEventDestroy(ctx->copy_ev);
    for (hipEvent_t e : ctx->pipe_ev) hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) {
        if (ctx->fwd_ev[i]) hipEventDestroy(ctx->fwd_ev[i]);
        if (ctx->match_ev[i]) hipEventDestroy(ctx->match_ev[i]);
    }
    if (ctx->join_ev) hipEventDestroy(ctx->join_ev);
    if (ctx->match_stream) { hipStreamSynchronize(ctx->match_stream); hipStreamDestroy(ctx->match_stream); }
    if (ctx->copy_stream) { hipStreamSynchronize(ctx->copy_stream); hipStreamDestroy(ctx->copy_stream); }
    for (auto& kv : ctx->ws) hipFree(kv.second.first);
    for (auto& kv : ctx->split_w) hipFree(kv.second);
    for (auto& kv : ctx->pinned) hipHostFree(kv.second.first);
    if (ctx->se18.blob) hipFree(ctx->se18.blob);
    if (ctx->se18.blob16) hipFree(ctx->se18.blob16);
    if (ctx->se18.stem_w16) hipFree(ctx->se18.stem_w16);
    if (ctx->se18.stem_w16s) hipFree(ctx->se18.stem_w16s);
    for (int i = 0; i < 2; ++i)
        if (ctx->se18.l1_conv2_w16s[i]) hipFree(ctx->se18.l1_conv2_w16s[i]);
    if (ctx->se18.zero_page) hipFree(ctx->se18.zero_page);
    if (ctx->se18.ep) hipFree(ctx->se18.ep);
    for (auto& e : ctx->ev_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto& p : ctx->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    hipEventDestroy(ctx->t0);
    hipEventDestroy(ctx->t1);
    hipStreamDestroy(ctx->own_stream);
    if (ctx->fault) hipHostFree(ctx->fault);
    delete ctx;
    return REID_OK;
}

extern "C" int reid_ctx_set_stream(reid_ctx* ctx, void* s) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    return REID_OK;
}

// The HIP null ("legacy default") stream: what torch.cuda.current_stream() is unless the host changed it; its handle is 0, which
// reid_ctx_set_stream reads as "back to the context's own stream", hence a separate entry point.
extern "C" int reid_ctx_set_null_stream(reid_ctx* ctx) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    ctx->stream = nullptr;
    return REID_OK;
}

// The sticky fault word (reid_internal.h): set by kernels, read here.  Every entry point that starts work calls this first
// (CTX_ENTER) and the synchronising ones once more after their wait, so the call that produced a fault - or the next one - fails.
int ctx_fault_status(reid_ctx* ctx) {
    if (!ctx->fault) return REID_OK;
    const volatile int* f = ctx->fault;
    if (f[0]) {
        reid_set_error("fp32-class arithmetic (reid_ctx_set_precision 2): an activation outside f16's range (|x| >= 65504, or NaN) reached a "
                       "split-operand site; results since the last reid_ctx_clear_fault are invalid - run this checkpoint in mode 0");
        return REID_ERR_STATE;
    }
    if (f[1]) {
        reid_set_error("a non-finite embedding left the neck (overflow upstream); results since the last reid_ctx_clear_fault are invalid");
        return REID_ERR_STATE;
    }
    if (f[2]) {
        reid_set_error("a split-K rendezvous of a convolution did not complete within its bound (conv3x3_x3.hip: the blocks of an output tile wait "
                       "for each other); results since the last reid_ctx_clear_fault are invalid");
        return REID_ERR_STATE;
    }
    return REID_OK;
}

extern "C" int reid_ctx_clear_fault(reid_ctx* ctx) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->fault) ctx->fault[0] = ctx->fault[1] = ctx->fault[2] = 0;
    return REID_OK;
}

extern "C" int reid_ctx_sync(reid_ctx* ctx) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->match_stream) HIP_TRY(hipStreamSynchronize(ctx->match_stream));
    return ctx_fault_status(ctx);
}

// every stream of the context's device (bench.py's timing bracket: barrier + device-wide synchronise)
extern "C" int reid_device_sync(reid_ctx* ctx) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    HIP_TRY(hipDeviceSynchronize());
    return ctx_fault_status(ctx);
}

extern "C" int reid_ctx_set_chunk(reid_ctx* ctx, int n) {
    ARG_CHECK(ctx && n >= 1 && n <= 4096);
    CTX_GUARD(ctx);
    ctx->chunk = n;
    return REID_OK;
}

extern "C" int reid_ctx_set_precision(reid_ctx* ctx, int mode) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    if (mode != 0 && mode != 1 && mode != 2) {
        reid_set_error("reid_ctx_set_precision: mode must be 0 (exact fp32 MFMA), 1 (fp16 storage / fp32 accumulate) or "
                       "2 (fp32-class: hi/lo-split operands on the f16 matrix pipe, fp32 storage)");
        return REID_ERR_ARG;
    }
    if (mode == 2) {   // the split form [wh 2^11 | wh | wl'] needs |w| 2^11 inside f16: a checkpoint outside it stays on modes 0 / 1
        const std::string& bad = !ctx->split_bad_se18.empty() ? ctx->split_bad_se18 : ctx->split_bad_swin;
        if (!bad.empty()) {
            reid_set_error("reid_ctx_set_precision(2): weight tensor %s of the loaded checkpoint is outside the range the fp32-class "
                           "arithmetic can split (|w| 2^11 < 65504); use mode 0", bad.c_str());
            return REID_ERR_ARG;
        }
    }
    ctx->precision = mode;
    return REID_OK;
}

extern "C" int reid_ctx_precision_ok(reid_ctx* ctx, int arch, int mode) {
    ARG_CHECK(ctx && (arch == 0 || arch == 1) && mode >= 0 && mode <= 2);
    const std::string& bad = arch == 0 ? ctx->split_bad_se18 : ctx->split_bad_swin;
    if (mode == 2 && !bad.empty()) {
        reid_set_error("reid_ctx_precision_ok: weight tensor %s of the loaded %s checkpoint is outside the range the fp32-class arithmetic can "
                       "split (|w| 2^11 < 65504)", bad.c_str(), arch == 0 ? "ResNet18-IBN" : "Swin");
        return REID_ERR_ARG;
    }
    return REID_OK;
}

extern "C" int reid_ctx_fault_peek(reid_ctx* ctx, int* bits) {
    ARG_CHECK(ctx && bits);
    const volatile int* f = ctx->fault;
    *bits = f ? ((f[0] ? 1 : 0) | (f[1] ? 2 : 0) | (f[2] ? 4 : 0)) : 0;
    return REID_OK;
}

extern "C" int reid_ctx_set_side_index(reid_ctx* ctx, const int32_t* index, int n) {
    ARG_CHECK(ctx && n >= 0 && (index || n == 0));
    CTX_GUARD(ctx);
    for (int i = 0; i < n; ++i) ARG_CHECK(index[i] >= 0);
    ctx->side_idx.assign(index, index + n);
    ctx->side_cursor = 0;
    return REID_OK;
}

int ctx_take_side(reid_ctx* ctx, int n, int rows, const char* what, const int32_t** d_idx) {
    *d_idx = nullptr;
    if (ctx->side_idx.empty()) return REID_OK;
    if (rows <= 0) {
        reid_set_error("%s: side indices are pending (reid_ctx_set_side_index) but the loaded weights carry no table for them", what);
        ctx->side_idx.clear();
        return REID_ERR_STATE;
    }
    if (ctx->side_cursor + (size_t)n > ctx->side_idx.size()) {
        reid_set_error("%s: %d images but only %zu side indices are pending", what, n, ctx->side_idx.size() - ctx->side_cursor);
        ctx->side_idx.clear();
        return REID_ERR_ARG;
    }
    const int32_t* h = ctx->side_idx.data() + ctx->side_cursor;
    for (int i = 0; i < n; ++i)
        if (h[i] >= rows) {
            reid_set_error("%s: side index %d of image %d is outside the table of %d rows", what, h[i], i, rows);
            ctx->side_idx.clear();
            return REID_ERR_ARG;
        }
    int32_t* d;
    REID_TRY(ctx_ws(ctx, "side.idx", (size_t)n * 4, (void**)&d));
    HIP_TRY(hipMemcpyAsync(d, h, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));      // h lives in a vector the next set call may reallocate
    ctx->side_cursor += n;
    if (ctx->side_cursor == ctx->side_idx.size()) ctx->side_idx.clear();
    *d_idx = d;
    return REID_OK;
}

extern "C" int reid_ctx_set_debug_keep(reid_ctx* ctx, int on) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    ctx->debug_keep = on < 0 ? 0 : (on > 2 ? 2 : on);   // 1: every kernel unfused, 2: production kernels, stage 0 not produced
    return REID_OK;
}

int ctx_ws(reid_ctx* ctx, const char* name, size_t bytes, void** out) {
    auto it = ctx->ws.find(name);
    if (it != ctx->ws.end() && it->second.second >= bytes) {
        *out = it->second.first;
        return REID_OK;
    }
    bool regrow = false;
    if (it != ctx->ws.end()) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipFree(it->second.first));
        ctx->ws.erase(it);
        regrow = true;
    }
    void* p = nullptr;
    // A buffer that had to grow once will grow again (tracking frames: every new maximum of the detection count re-allocated
    // ~10 buffers behind a stream synchronisation - 13 % of a 600-frame run): half as much again on top, up to 1 GiB extra
    size_t cap = bytes < 256 ? 256 : bytes;
    if (regrow) cap += cap / 2 < ((size_t)1 << 30) ? cap / 2 : ((size_t)1 << 30);
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) {
        reid_set_error("hipMalloc(%zu) for workspace '%s' failed: %s", cap, name, hipGetErrorString(e));
        return REID_ERR_NOMEM;
    }
    ctx->ws[name] = {p, cap};
    *out = p;
    return REID_OK;
}

// grow-only named PINNED host buffer (staging of the frame pipeline: copies from / to it are real asynchronous DMAs, whereas a
// hipMemcpyAsync on pageable memory blocks the caller until everything queued before it has run)
int ctx_pinned(reid_ctx* ctx, const char* name, size_t bytes, void** out) {
    auto it = ctx->pinned.find(name);
    if (it != ctx->pinned.end() && it->second.second >= bytes) {
        *out = it->second.first;
        return REID_OK;
    }
    if (it != ctx->pinned.end()) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipHostFree(it->second.first));
        ctx->pinned.erase(it);
    }
    void* p = nullptr;
    size_t cap = bytes < 4096 ? 4096 : bytes + bytes / 2;
    hipError_t e = hipHostMalloc(&p, cap, hipHostMallocDefault);
    if (e != hipSuccess) {
        reid_set_error("hipHostMalloc(%zu) for staging buffer '%s' failed: %s", cap, name, hipGetErrorString(e));
        return REID_ERR_NOMEM;
    }
    ctx->pinned[name] = {p, cap};
    *out = p;
    return REID_OK;
}

extern "C" int reid_host_alloc(reid_ctx* ctx, size_t bytes, void** out) {
    ARG_CHECK(ctx && out && bytes > 0);
    CTX_GUARD(ctx);
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        reid_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return REID_ERR_NOMEM;
    }
    return REID_OK;
}

extern "C" int reid_host_free(reid_ctx* ctx, void* p) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    if (p) HIP_TRY(hipHostFree(p));
    return REID_OK;
}

extern "C" int reid_malloc(reid_ctx* ctx, size_t bytes, void** dptr) {
    ARG_CHECK(ctx && dptr);
    CTX_GUARD(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        reid_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return REID_ERR_NOMEM;
    }
    return REID_OK;
}
extern "C" int reid_free(reid_ctx* ctx, void* dptr) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(dptr));
    return REID_OK;
}
extern "C" int reid_memcpy_h2d(reid_ctx* ctx, void* dst, const void* src, size_t bytes) {
    ARG_CHECK(ctx && dst && src);
    CTX_GUARD(ctx);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}
extern "C" int reid_memcpy_d2h(reid_ctx* ctx, void* dst, const void* src, size_t bytes) {
    ARG_CHECK(ctx && dst && src);
    CTX_GUARD(ctx);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}

// ------------------------------------------------------------------------------------------------ timing
extern "C" int reid_timer_start(reid_ctx* ctx) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    HIP_TRY(hipEventRecord(ctx->t0, ctx->stream));
    return REID_OK;
}
extern "C" int reid_timer_stop(reid_ctx* ctx, float* ms) {
    ARG_CHECK(ctx && ms);
    CTX_GUARD(ctx);
    HIP_TRY(hipEventRecord(ctx->t1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->t1));
    HIP_TRY(hipEventElapsedTime(ms, ctx->t0, ctx->t1));
    return REID_OK;
}

void prof_begin(reid_ctx* ctx, int kind, double flops, double bytes) {
    if (!ctx->profile) return;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!ctx->ev_pool.empty()) {
        ev = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
    } else {
        hipEventCreate(&ev.first);
        hipEventCreate(&ev.second);
    }
    hipEventRecord(ev.first, ctx->stream);
    ctx->pending.push_back({kind, ev.first, ev.second, flops, bytes});
}
void prof_end(reid_ctx* ctx) {
    if (!ctx->profile || ctx->pending.empty()) return;
    hipEventRecord(ctx->pending.back().b, ctx->stream);
}
static int prof_drain(reid_ctx* ctx) {
    if (ctx->pending.empty()) return REID_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto& p : ctx->pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        ProfSlot& s = ctx->prof[p.kind];
        s.ms += ms;
        s.flops += p.flops;
        s.bytes += p.bytes;
        s.launches += 1;
        ctx->ev_pool.push_back({p.a, p.b});
    }
    ctx->pending.clear();
    return REID_OK;
}
extern "C" int reid_profile_enable(reid_ctx* ctx, int on) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    REID_TRY(prof_drain(ctx));
    ctx->profile = on != 0;
    return REID_OK;
}
extern "C" int reid_profile_reset(reid_ctx* ctx) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    REID_TRY(prof_drain(ctx));
    for (auto& s : ctx->prof) s = ProfSlot();
    return REID_OK;
}
extern "C" int reid_profile_get(reid_ctx* ctx, int kind, double* ms, long long* launches, double* flops, double* bytes) {
    ARG_CHECK(ctx && kind >= 0 && kind < REID_K_COUNT);
    CTX_GUARD(ctx);
    REID_TRY(prof_drain(ctx));
    if (ms) *ms = ctx->prof[kind].ms;
    if (launches) *launches = ctx->prof[kind].launches;
    if (flops) *flops = ctx->prof[kind].flops;
    if (bytes) *bytes = ctx->prof[kind].bytes;
    return REID_OK;
}

// ------------------------------------------------------------------------------------------------ weights
static const int kBlkC[8] = {64, 64, 128, 128, 256, 256, 512, 512};
static const int kBlkCin[8] = {64, 64, 64, 128, 128, 256, 256, 512};
static const int kBlkStride[8] = {1, 1, 2, 1, 2, 1, 1, 1};  // block41: last stride forced to 1 (SERes18_IBN.py:99-101)
static const int kBlkIbn[8] = {1, 1, 1, 1, 1, 1, 0, 0};
static const int kBlkDs[8] = {0, 0, 1, 0, 1, 0, 1, 0};
static const char* kBlkName[8] = {"b11", "b12", "b21", "b22", "b31", "b32", "b41", "b42"};

extern "C" int reid_seres18_load(reid_ctx* ctx, const float* blob, size_t n_floats, const char* manifest) {
    ARG_CHECK(ctx && blob && manifest && n_floats > 0);
    CTX_GUARD(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    std::map<std::string, std::pair<size_t, size_t>> tab;
    {
        std::istringstream in(manifest);
        std::string name;
        size_t off, cnt;
        while (in >> name >> off >> cnt) {
            if (off + cnt > n_floats) {
                reid_set_error("manifest entry '%s' [%zu,+%zu) exceeds blob of %zu floats", name.c_str(), off, cnt, n_floats);
                return REID_ERR_ARG;
            }
            if (off % 4 != 0) {
                reid_set_error("manifest entry '%s' is not 16-byte aligned", name.c_str());
                return REID_ERR_ARG;
            }
            tab[name] = {off, cnt};
        }
    }
    // precision 2 operand range (include/reid_hip.h): convolution weights are split as [wh 2^11 | wh | wl'] f16, the stem's as
    // [wh | wl' | wll']; reid_model_factory.load_pretrained_weights accepts any shape-compatible checkpoint
    // (modification_tracking/reid_model_factory.py:158-210), so the range is checked here, on the host copy
    const std::string bad = split_range_violation(blob, tab, {{".conv1.w", 65504.0f / 2048.0f}, {".conv2.w", 65504.0f / 2048.0f},
                                                              {".ds.w", 65504.0f / 2048.0f}, {"stem.w", 65504.0f}});
    if (!bad.empty() && ctx->precision == 2) {
        reid_set_error("reid_seres18_load: weight tensor %s cannot be split for the fp32-class arithmetic selected on this context "
                       "(reid_ctx_set_precision 2 needs |w| 2^11 < 65504); load it in mode 0", bad.c_str());
        return REID_ERR_ARG;
    }
    ctx->split_bad_se18 = bad;
    Se18Weights& w = ctx->se18;
    if (w.blob) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipFree(w.blob));
        if (w.blob16) HIP_TRY(hipFree(w.blob16));
        if (w.stem_w16) HIP_TRY(hipFree(w.stem_w16));
        if (w.stem_w16s) HIP_TRY(hipFree(w.stem_w16s));
        for (int i = 0; i < 2; ++i)
            if (w.l1_conv2_w16s[i]) HIP_TRY(hipFree(w.l1_conv2_w16s[i]));
        if (w.zero_page) HIP_TRY(hipFree(w.zero_page));
        if (w.ep) HIP_TRY(hipFree(w.ep));
        w = Se18Weights();
        for (auto& kv : ctx->split_w) (void)hipFree(kv.second);   // split forms of the old blob's weights
        ctx->split_w.clear();
    }
    HIP_TRY(hipMalloc((void**)&w.blob, n_floats * sizeof(float)));
    HIP_TRY(hipMemcpy(w.blob, blob, n_floats * sizeof(float), hipMemcpyHostToDevice));
    w.n_floats = n_floats;
    bool missing = false;
    std::string first_missing;
    auto get = [&](const std::string& name, size_t expect) -> const float* {
        auto it = tab.find(name);
        if (it == tab.end() || (expect && it->second.second != expect)) {
            if (!missing) first_missing = name;
            missing = true;
            return nullptr;
        }
        return w.blob + it->second.first;
    };
    w.arch = tab.count("b11.ta") ? 1 : (tab.count("b11.ema") ? 2 : 0);   // sibling backbones carry their attention tensors instead of se.*
    w.stem_w = get("stem.w", 64 * 192);
    w.stem_scale = get("stem.scale", 64);
    w.stem_shift = get("stem.shift", 64);
    for (int i = 0; i < 8; ++i) {
        Se18Block& b = w.blk[i];
        const std::string n = kBlkName[i];
        b.c = kBlkC[i]; b.cin = kBlkCin[i]; b.stride = kBlkStride[i]; b.ibn = kBlkIbn[i]; b.ds = kBlkDs[i];
        b.mid = b.c / 16 > 8 ? b.c / 16 : 8;  // SERes18_IBN.py:17
        const int half = b.ibn ? b.c / 2 : 0;
        b.conv1_w = get(n + ".conv1.w", (size_t)b.c * 9 * b.cin);
        b.in_gamma = b.ibn ? get(n + ".n1.in_gamma", half) : nullptr;
        b.in_beta = b.ibn ? get(n + ".n1.in_beta", half) : nullptr;
        b.bn1_scale = get(n + ".n1.bn_scale", b.c - half);
        b.bn1_shift = get(n + ".n1.bn_shift", b.c - half);
        b.conv2_w = get(n + ".conv2.w", (size_t)b.c * 9 * b.c);
        b.bn2_scale = get(n + ".bn2.scale", b.c);
        b.bn2_shift = get(n + ".bn2.shift", b.c);
        b.ds_w = b.ds ? get(n + ".ds.w", (size_t)b.c * b.cin) : nullptr;
        b.ds_scale = b.ds ? get(n + ".ds.scale", b.c) : nullptr;
        b.ds_shift = b.ds ? get(n + ".ds.shift", b.c) : nullptr;
        b.se_w1 = b.se_w2 = b.ta = b.ema = nullptr;
        if (w.arch == 0) {
            b.se_w1 = get(n + ".se.w1", (size_t)b.mid * b.c);
            b.se_w2 = get(n + ".se.w2t", (size_t)b.c * b.mid);   // [mid][C]
        } else if (w.arch == 1) {
            b.ta = get(n + ".ta", 300);
        } else {
            const size_t cg = b.c / 32;
            b.ema = get(n + ".ema", cg * cg + cg + cg * cg * 9 + cg + 2 * cg);
        }
    }
    w.gem_p = get("gem.p", 1);
    w.neck_scale = get("neck.scale", 512);
    w.neck_shift = get("neck.shift", 512);
    auto cls = tab.find("cls.w");
    if (cls != tab.end() && cls->second.second % 512 == 0) {
        w.cls_w = w.blob + cls->second.first;
        w.num_class = (int)(cls->second.second / 512);
    } else {
        w.cls_w = nullptr;
        w.num_class = 0;
    }
    // optional: the camera-bias table of SERse18_IBN.forward(x, cam) (SERes18_IBN.py:246-248, 269-270)
    auto cb = tab.find("cam.bias");
    auto cf = tab.find("cam.factor");
    w.cam_bias = nullptr;
    w.num_cams = 0;
    if (cb != tab.end() && cf != tab.end() && cb->second.second % 512 == 0 && cf->second.second == 1) {
        w.cam_bias = w.blob + cb->second.first;
        w.num_cams = (int)(cb->second.second / 512);
        w.cam_factor = blob[cf->second.first];
    }
    if (missing) {
        reid_set_error("reid_seres18_load: manifest entry '%s' missing or of unexpected size", first_missing.c_str());
        HIP_TRY(hipFree(w.blob));
        w = Se18Weights();
        return REID_ERR_ARG;
    }
    {   // conv1 epilogue vectors of the IBN blocks: identity on the InstanceNorm half, folded BatchNorm on the rest
        std::vector<float> ep(8 * 1024, 0.f);
        for (int i = 0; i < 8; ++i) {
            const Se18Block& b = w.blk[i];
            const int half = b.ibn ? b.c / 2 : 0;
            for (int ch = 0; ch < b.c; ++ch) {
                ep[i * 1024 + ch] = ch < half ? 1.f : blob[(b.bn1_scale - w.blob) + (ch - half)];
                ep[i * 1024 + 512 + ch] = ch < half ? 0.f : blob[(b.bn1_shift - w.blob) + (ch - half)];
            }
        }
        HIP_TRY(hipMalloc((void**)&w.ep, ep.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(w.ep, ep.data(), ep.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // fp16 copies for the fp16 path: whole blob at the same element offsets + the padded-NHWC4 stem weights
    HIP_TRY(hipMalloc((void**)&w.blob16, n_floats * sizeof(_Float16)));
    HIP_TRY(hipMalloc((void**)&w.stem_w16, 64 * 256 * sizeof(_Float16)));
    HIP_TRY(hipMalloc((void**)&w.stem_w16s, 64 * 256 * sizeof(_Float16)));
    HIP_TRY(hipMalloc((void**)&w.zero_page, 256));
    HIP_TRY(hipMemsetAsync(w.zero_page, 0, 256, ctx->stream));
    REID_TRY(launch_f32_to_f16(ctx, w.blob, n_floats, w.blob16));
    REID_TRY(launch_stem_w16(ctx, w.stem_w, w.stem_w16));
    REID_TRY(launch_stem_w16_scaled(ctx, w.stem_w, w.stem_scale, w.stem_w16s));
    for (int i = 0; i < 2; ++i) {   // layer-1 conv2 weights with the BN scale folded in (conv3x3_c64_f16.hip)
        HIP_TRY(hipMalloc((void**)&w.l1_conv2_w16s[i], 64 * 576 * sizeof(_Float16)));
        REID_TRY(launch_scale_rows_f16(ctx, w.blk[i].conv2_w, w.blk[i].bn2_scale, 64, 576, w.l1_conv2_w16s[i]));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    w.loaded = true;
    return REID_OK;
}

extern "C" int reid_seres18_dims(reid_ctx* ctx, int* embed_dim, int* num_class) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    if (!ctx->se18.loaded) {
        reid_set_error("no weights loaded");
        return REID_ERR_STATE;
    }
    if (embed_dim) *embed_dim = 512;
    if (num_class) *num_class = ctx->se18.num_class;
    return REID_OK;
}

// ------------------------------------------------------------------------------------------------ forward
static const int IMG_H = 256, IMG_W = 128;

// precision 2: does this convolution run in split arithmetic (conv_gemm below), i.e. read the packed [xh | xl'] input only?
// 3x3 stride-1 convolutions always do (LDS-halo kernel); strided / 1x1 ones when there are enough 128-wide tiles for the SPLIT GEMM
// (no split-K form) - a tracking-sized batch leaves them on the exact-fp32 kernel, which splits K for small launches.
static bool conv_split_path(reid_ctx* ctx, int n, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad) {
    if (ctx->precision != 2 || Cin % 64 != 0 || Cout % 64 != 0 || !ctx->se18.zero_page) return false;
    Gemm16Params q;
    memset(&q, 0, sizeof(q));
    q.H = H; q.W = W; q.Cin = ctx->split_terms * Cin; q.R = R; q.S = S; q.stride = stride; q.pad = pad;
    q.Ho = (H + 2 * pad - R) / stride + 1;
    q.Wo = (W + 2 * pad - S) / stride + 1;
    q.M = n * q.Ho * q.Wo; q.N = Cout;
    const bool halo = R == 3 && S == 3 && stride == 1 && pad == 1 && conv3x3_f16_supported(q);
    const bool enough = halo || (Cout % 128 == 0 && (long long)(q.M / 256) * (Cout / 128) >= (R == 3 ? ctx->split_gemm_min_tiles * 3 / 4 : ctx->split_gemm_min_tiles));   // (layer 3's strided 3x3 at 120 crops: 107 -> 70 us; at 64 crops 49 against 67: stays)
    if (q.M % 128 == 0 && !enough && !halo && ctx->conv_x3s == 2) {      // experiment (switch conv_x3s = 2): the small launches too on conv_x3s_kernel, so
        q.split_terms = ctx->split_terms; q.ldb = (long long)R * S * q.Cin; q.ldc = Cout;   // that a layer's arithmetic does not depend on the batch size - a 30-crop
        return conv_x3s_supported(ctx, q);                                                 // frame pays 8 us for it (1x1: 13-20 us against 7-15 in exact fp32)
    }
    return q.M % 128 == 0 && enough;
}

int conv_gemm(reid_ctx* ctx, int amode, const void* x, int n, int H, int W, int Cin, const float* wgt, int Cout, int R,
                     int S, int stride, int pad, int Kpad, const float* a_scale, const float* a_shift, int a_relu,
                     const float* col_scale, const float* col_shift, const float* residual, int relu, float* stats,
                     float* out, int relu_from, const _Float16* x_packed, _Float16* out_packed, int pack_from, bool* packed_written) {
    if (packed_written) *packed_written = false;
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = x;
    p.H = H; p.W = W; p.Cin = Cin; p.R = R; p.S = S; p.stride = stride; p.pad_y = pad; p.pad_x = pad;
    p.Ho = (H + 2 * pad - R) / stride + 1;
    p.Wo = (W + 2 * pad - S) / stride + 1;
    p.a_scale = a_scale; p.a_shift = a_shift; p.a_relu = a_relu;
    p.B = wgt; p.ldb = Kpad;
    p.M = n * p.Ho * p.Wo; p.N = Cout; p.K = Kpad;
    p.C = out; p.ldc = Cout;
    p.col_scale = col_scale; p.col_shift = col_shift; p.residual = residual; p.relu = relu; p.stats = stats;
    p.relu_from = relu_from;
    p.diag = ctx->conv_diag;
    const double ktrue = (double)R * S * Cin;
    const double flops = 2.0 * p.M * Cout * ktrue;
    const double in_bytes = (double)n * H * W * Cin * (amode == A_STEM_U8 ? 1.0 : 4.0);
    const double bytes = in_bytes + ((double)p.M * Cout + (double)Cout * ktrue) * 4.0 + (residual ? (double)p.M * Cout * 4.0 : 0.0);
    if (amode == A_IM2COL && !a_scale && conv_split_path(ctx, n, H, W, Cin, Cout, R, S, stride, pad)) {
        // "fp32-class" arithmetic on the f16 matrix pipe (Gemm16Params, SPLIT builds of conv3x3_f16.hip / gemm_f16.hip): the fp32
        // activations are packed to [xh | xl'] f16, the weights were split once; three f16 products per multiply, fp32 accumulate,
        // fp32 in / out - the layers around the convolution (IBN, SE, residual stream) are the exact-fp32 path's, untouched
        Gemm16Params q;
        memset(&q, 0, sizeof(q));
        const int T = ctx->split_terms;
        q.split_terms = T;
        q.H = H; q.W = W; q.Cin = T * Cin; q.R = R; q.S = S; q.stride = stride; q.pad = pad;
        q.Ho = p.Ho; q.Wo = p.Wo;
        q.M = p.M; q.N = Cout; q.K = R * S * T * Cin; q.ldb = q.K;
        const bool halo = R == 3 && S == 3 && stride == 1 && pad == 1 && conv3x3_f16_supported(q);
        {
            const _Float16* a16 = x_packed;
            if (!a16) {
                _Float16* buf;
                const long long rows_in = (long long)n * H * W;
                REID_TRY(ctx_ws(ctx, "split.a", (size_t)rows_in * 2 * Cin * 2, (void**)&buf));
                REID_TRY(launch_split_pack(ctx, (const float*)x, rows_in, Cin, buf));
                a16 = buf;
            }
            auto it = ctx->split_w.find(wgt);
            if (it == ctx->split_w.end()) {
                void* w16;
                HIP_TRY(hipMalloc(&w16, (size_t)Cout * q.K * 2));
                REID_TRY(launch_split_weights(ctx, wgt, Cout, R * S, Cin, T, (_Float16*)w16));
                it = ctx->split_w.emplace(wgt, w16).first;
            }
            q.A = a16;
            q.B = (const _Float16*)it->second;
            q.C32 = out; q.ldc = Cout;
            q.col_scale = col_scale; q.col_shift = col_shift; q.res32 = residual; q.relu = relu; q.relu_from = relu_from;
            q.stats = stats;
            q.acc_scale = 1.0f / 2048.0f;
            q.zero_page = ctx->se18.zero_page;
            if (out_packed && packed_written && pack_from % 32 == 0) {   // the consumer reads [yh | yl']: written by this epilogue
                q.pack16 = out_packed;
                q.pack_from = pack_from;
                *packed_written = true;
            }
            if (halo) return launch_conv3x3_split(ctx, q, REID_K_CONV_GEMM, flops, bytes);
            // strided 3x3 / 1x1: conv_x3s_kernel (round 6) wherever gemm_f16.hip's SPLIT build served - 1024 crops: 440 / 109 / 349 / 76 / 219 ->
            // 333 / 103 / 246 / 60 / 156 us for the five launches of a pass, 120 crops: 185 -> 180 us in total
            if (conv_x3s_supported(ctx, q)) return launch_conv_x3s(ctx, q, REID_K_CONV_GEMM, flops, bytes);
            return launch_gemm_f16_split(ctx, q, REID_K_CONV_GEMM, flops, bytes);
        }
    }
    if (amode == A_IM2COL && ctx->f32_conv && conv_f32_supported(p)) return launch_conv_f32(ctx, p, REID_K_CONV_GEMM, flops, bytes);
    return launch_gemm_f32(ctx, amode, E_CONV, p, REID_K_CONV_GEMM, flops, bytes);
}

// precision 2: the [wh 2^11 | wh | wl'] f16 form of a convolution's weights (made once per checkpoint, cached by the blob address)
static int split_weights_of(reid_ctx* ctx, const float* wgt, int Cout, int taps, int Cin, const _Float16** out) {
    auto it = ctx->split_w.find(wgt);
    if (it == ctx->split_w.end()) {
        void* w16;
        HIP_TRY(hipMalloc(&w16, (size_t)Cout * taps * ctx->split_terms * Cin * 2));
        REID_TRY(launch_split_weights(ctx, wgt, Cout, taps, Cin, ctx->split_terms, (_Float16*)w16));
        it = ctx->split_w.emplace(wgt, w16).first;
    }
    *out = (const _Float16*)it->second;
    return REID_OK;
}

#ifdef REID_EXPERIMENTS
// Layer 4 of a small batch as ONE chain launch (conv3x3_x3.hip chain_kernel): conv1 / conv2 / SE tail of blocks 41 and 42, six stages,
// items ordered stage by stage and image pair by image pair; block 41's 1x1 shortcut convolution was launched before (it reads the
// layer input only).  Reference: SERes18_IBN.py:96-128 (SEBasicBlock), :223-226 (layer 4: stride forced to 1).
static int seres18_chain_layer4(reid_ctx* ctx, int n, const Se18Block& ka, const Se18Block& kb, const _Float16* in16, const float* sc,
                                float* y, float* out_a, float* out, float* stats) {
    const int H = 16, W = 8, hw = 128, C = 512, SK = 4, nnt = C / 128;
    const int nmt = (n * hw + 255) / 256;
    _Float16 *c1_16, *a16;
    float* skws;
    int* cnt;
    REID_TRY(ctx_ws(ctx, "split.c1", (size_t)n * hw * C * 2 * 2, (void**)&c1_16));
    REID_TRY(ctx_ws(ctx, "chain.a16", (size_t)n * hw * C * 2 * 2, (void**)&a16));
    const size_t part = (size_t)nmt * nnt * SK * 256 * 128;          // floats of one convolution's split-K partials
    REID_TRY(ctx_ws(ctx, "chain.splitk", 4 * part * sizeof(float), (void**)&skws));
    const bool fresh = ctx->ws.find("chain.cnt") == ctx->ws.end();
    REID_TRY(ctx_ws(ctx, "chain.cnt", (4 * 1024 + 64 + 8 * 64) * sizeof(int), (void**)&cnt));
    if (fresh) HIP_TRY(hipMemsetAsync(cnt, 0, (4 * 1024 + 64 + 8 * 64) * sizeof(int), ctx->stream));
    ChainParams cp;
    memset(&cp, 0, sizeof(cp));
    cp.n_img = n;
    cp.flags = ctx->chain;
    cp.counters = cnt + 4 * 1024;
    cp.fault = ctx->fault;
    auto conv = [&](int idx, const _Float16* a, int cin, const float* wgt, const float* scale, const float* shift, const float* res, int relu,
                    float* stats_out, _Float16* pack) -> int {
        Gemm16Params& q = cp.conv[idx];
        q.H = H; q.W = W; q.Cin = 3 * cin; q.R = 3; q.S = 3; q.stride = 1; q.pad = 1; q.Ho = H; q.Wo = W;
        q.M = n * hw; q.N = C; q.K = 9 * 3 * cin; q.ldb = q.K;
        q.split_terms = 3;
        q.A = a;
        REID_TRY(split_weights_of(ctx, wgt, C, 9, cin, &q.B));
        q.C32 = y; q.ldc = C;
        q.col_scale = scale; q.col_shift = shift; q.res32 = res; q.relu = relu; q.relu_from = 0;
        q.stats = stats_out;
        q.acc_scale = 1.0f / 2048.0f;
        q.zero_page = ctx->se18.zero_page;
        q.pack16 = pack; q.pack_from = 0;
        q.split_k = SK; q.splitk_ws = skws + (size_t)idx * part; q.splitk_cnt = cnt + idx * 1024;      // x3m16_tail: [tile] arrivals, [512 + tile] readers done
        q.fault = ctx->fault;
        return REID_OK;
    };
    REID_TRY(conv(0, in16, ka.cin, ka.conv1_w, ka.bn1_scale, ka.bn1_shift, nullptr, 1, nullptr, c1_16));
    REID_TRY(conv(1, c1_16, C, ka.conv2_w, ka.bn2_scale, ka.bn2_shift, nullptr, 0, stats, nullptr));
    REID_TRY(conv(2, a16, C, kb.conv1_w, kb.bn1_scale, kb.bn1_shift, nullptr, 1, nullptr, c1_16));
    REID_TRY(conv(3, c1_16, C, kb.conv2_w, kb.bn2_scale, kb.bn2_shift, out_a, 1, stats, nullptr));
    const int slices = 2;
    auto se = [&](int idx, const Se18Block& k, const float* shortcut, float* o, _Float16* pk) {
        ChainElem& e = cp.el[idx];
        e.x = y; e.stats = stats; e.sc = shortcut; e.g = k.se_w1; e.b = k.se_w2; e.out = o; e.packed = pk;
        e.c = C; e.half = 0; e.hw = hw; e.tiles = 1; e.mid = k.mid; e.slices = slices;
    };
    se(0, ka, sc, out_a, a16);
    se(1, kb, out_a, out, nullptr);
    const int conv_items = nmt * nnt * SK, se_items = n * slices;
    const int kinds[6] = {0, 0, 2, 0, 0, 2}, idxs[6] = {0, 1, 0, 2, 3, 1};
    const int targets[6] = {0, nnt * SK, nnt * SK, slices, nnt * SK, nnt * SK};
    int first = 0;
    for (int s = 0; s < 6; ++s) {
        ChainStage& st = cp.st[s];
        st.kind = kinds[s]; st.idx = idxs[s]; st.first = first; st.items = kinds[s] == 0 ? conv_items : se_items;
        st.dep = s - 1; st.target = targets[s]; st.sk = SK;
        first += st.items;
    }
    cp.n_stages = 6;
    cp.total_items = first;
    prof_begin(ctx, REID_K_CONV_GEMM, 2.0 * n * hw * C * 9.0 * (ka.cin + 3.0 * C), 0.0);
    const int st = launch_chain(ctx, cp, W);
    prof_end(ctx);
    return st;
}
#endif   // REID_EXPERIMENTS

struct Se18Bufs {
    float *stem, *pool, *t[4], *stats, *a_scale, *a_shift, *se, *gem;
    float* stage[11];
};

// x: uint8 NHWC crops (is_u8) or fp32 NHWC, both [n][256][128][3] on the device
static int seres18_forward(reid_ctx* ctx, const void* x, bool is_u8, int n, float* d_emb, float* d_logits) {
    Se18Weights& w = ctx->se18;
    if (!w.loaded) {
        reid_set_error("reid_embed_*: call reid_seres18_load first");
        return REID_ERR_STATE;
    }
    Se18Bufs b;
    const size_t per = 131072;  // largest block tensor per crop (64 x 32 x 64)
    const bool keep = ctx->debug_keep != 0;
    REID_TRY(ctx_ws(ctx, "se18.stem", (size_t)n * 524288 * 4, (void**)&b.stem));
    REID_TRY(ctx_ws(ctx, "se18.pool", (size_t)n * per * 4, (void**)&b.pool));
    const int nt = keep ? 1 : 4;
    float* tbase = nullptr;
    REID_TRY(ctx_ws(ctx, "se18.t", (size_t)n * per * 4 * (keep ? 4 * 8 : 4), (void**)&tbase));
    REID_TRY(ctx_ws(ctx, "se18.stats", (size_t)n * 2048 * 4, (void**)&b.stats));
    REID_TRY(ctx_ws(ctx, "se18.ascale", (size_t)n * 512 * 4, (void**)&b.a_scale));
    REID_TRY(ctx_ws(ctx, "se18.ashift", (size_t)n * 512 * 4, (void**)&b.a_shift));
    REID_TRY(ctx_ws(ctx, "se18.se", (size_t)n * 512 * 4, (void**)&b.se));
    REID_TRY(ctx_ws(ctx, "se18.gem", (size_t)n * 512 * 4, (void**)&b.gem));
    (void)nt;

    // stem: conv7x7 s2 p3 + BN, no ReLU (SERes18_IBN.py:251-253), then MaxPool2d(3,2,1) (:254)
    // stem_f32.hip: weights resident in LDS, A operand read from an fp32 LDS image of the input rows; the max-pool runs on its
    // accumulators unless the conv map itself is wanted (debug stage 0) or REID_F32_STEMPOOL=0
    const bool pool_fused = ctx->f32_conv == 1 && ctx->f32_stem_pool && ctx->debug_keep != 1;   // debug_keep 2: production kernels, no stage 0
    // precision 2: the block input as [xh | xl'] f16, written by the split stem (or packed from its fp32 output) and then by every
    // block's SE tail; conv1 and the shortcut's 1x1 convolution both read it
    _Float16* cur16 = nullptr;
    const bool split_mode = ctx->precision == 2 && w.arch == 0 && ctx->f32_conv == 1;
    if (split_mode) REID_TRY(ctx_ws(ctx, "split.cur", (size_t)n * per * 2 * 2, (void**)&cur16));
    const bool stem_split = split_mode && pool_fused && ctx->stem_split;
    if (stem_split) {
        REID_TRY(launch_stem_split(ctx, x, is_u8, n, w.stem_w, w.stem_scale, w.stem_shift, b.pool, cur16));
    } else if (pool_fused) {
        REID_TRY(launch_stem_f32(ctx, x, is_u8, n, w.stem_w, w.stem_scale, w.stem_shift, b.pool, true));
    } else if (ctx->f32_conv == 1) {
        REID_TRY(launch_stem_f32(ctx, x, is_u8, n, w.stem_w, w.stem_scale, w.stem_shift, b.stem, false));
    } else {
        REID_TRY(conv_gemm(ctx, is_u8 ? A_STEM_U8 : A_STEM_F32, x, n, IMG_H, IMG_W, 3, w.stem_w, 64, 7, 7, 2, 3, 192, nullptr,
                           nullptr, 0, w.stem_scale, w.stem_shift, nullptr, 0, nullptr, b.stem));
    }
    if (!pool_fused) REID_TRY(launch_maxpool3s2(ctx, b.stem, n, 128, 64, 64, b.pool));
    b.stage[0] = b.stem;
    b.stage[1] = b.pool;

    const float* cur = b.pool;
    int H = 64, W = 32;
    if (split_mode && !stem_split) REID_TRY(launch_split_pack(ctx, b.pool, (long long)n * 64 * 32, 64, cur16));
    for (int i = 0; i < 8; ++i) {
        const Se18Block& k = w.blk[i];
#ifdef REID_EXPERIMENTS
        if (i == 6 && (ctx->chain & 1) && cur16 && !keep && n <= 64 && ctx->pack_epilogue && ctx->split_terms == 3) {
            // layer 4 of a small batch: block 41's shortcut convolution, then ONE chain launch for everything else (chain_kernel)
            float* fr[3];
            int nf = 0;
            for (int j = 0; j < 4 && nf < 3; ++j)
                if (tbase + (size_t)j * n * per != cur) fr[nf++] = tbase + (size_t)j * n * per;
            float *sc = fr[0], *y = fr[1], *out_a = fr[2], *out = const_cast<float*>(cur);   // (the layer input's fp32 form is dead once the shortcut has read it)
            REID_TRY(conv_gemm(ctx, A_IM2COL, cur, n, H, W, k.cin, k.ds_w, k.c, 1, 1, k.stride, 0, k.cin, nullptr, nullptr, 0, k.ds_scale, k.ds_shift,
                               nullptr, 0, nullptr, sc, 0, cur16));
            REID_TRY(seres18_chain_layer4(ctx, n, k, w.blk[7], cur16, sc, y, out_a, out, b.stats));
            b.stage[8] = out_a;
            b.stage[9] = out;
            cur = out;
            break;
        }
#endif
        // four rotating buffers; in debug-keep mode every block gets its own four
        float* tb[4];
        for (int j = 0; j < 4; ++j) tb[j] = tbase + ((size_t)(keep ? i * 4 : 0) + j) * n * per;
        // pick buffers that do not alias the block input
        float* free_[3];
        int nf = 0;
        for (int j = 0; j < 4 && nf < 3; ++j)
            if (tb[j] != cur) free_[nf++] = tb[j];
        float* c1 = free_[0];
        float* y = free_[1];
        float* sc = free_[2];
        const int Ho = (H + 2 - 3) / k.stride + 1, Wo = (W + 2 - 3) / k.stride + 1;
        const int hw = Ho * Wo, tiles = hw / 128;
        const int half = k.ibn ? k.c / 2 : 0;
        _Float16* c1_16 = nullptr;
        if (ctx->f32_conv == 1) {
            // LDS-DMA conv kernel (conv_f32.hip): its loader copies, so bn1 is finished by the producer - the BatchNorm channels
            // (+ ReLU) in conv1's epilogue, the InstanceNorm half (statistics of the whole image) by one in-place pass
            // precision 2: conv2 (c1's only reader) loads [xh | xl'] - conv1's epilogue writes that form itself for the columns that
            // are finished there (the BatchNorm half of an IBN layer, everything in layer 4), the InstanceNorm half follows in one
            // pass (in_apply_pack); a conv1 that fell back to the fp32 kernel (strided, small launch) leaves fp32 only
            bool packed_c1 = false;
            if (cur16 && !keep && ctx->pack_epilogue) REID_TRY(ctx_ws(ctx, "split.c1", (size_t)n * hw * k.c * 2 * 2, (void**)&c1_16));
            if (k.ibn) {
                REID_TRY(conv_gemm(ctx, A_IM2COL, cur, n, H, W, k.cin, k.conv1_w, k.c, 3, 3, k.stride, 1, 9 * k.cin, nullptr, nullptr,
                                   0, w.ep + (size_t)i * 1024, w.ep + (size_t)i * 1024 + 512, nullptr, 1, b.stats, c1, half, cur16,
                                   c1_16, half, &packed_c1));
                if (cur16 && !keep) {   // InstanceNorm finish + [xh | xl'] in one pass
                    if (!c1_16) REID_TRY(ctx_ws(ctx, "split.c1", (size_t)n * hw * k.c * 2 * 2, (void**)&c1_16));
                    REID_TRY(launch_in_apply_pack(ctx, c1, b.stats, n, tiles, k.c, half, hw, k.in_gamma, k.in_beta, c1_16, packed_c1));
                } else {
                    REID_TRY(launch_in_apply(ctx, c1, b.stats, n, tiles, k.c, half, hw, k.in_gamma, k.in_beta));
                }
            } else {
                REID_TRY(conv_gemm(ctx, A_IM2COL, cur, n, H, W, k.cin, k.conv1_w, k.c, 3, 3, k.stride, 1, 9 * k.cin, nullptr, nullptr,
                                   0, k.bn1_scale, k.bn1_shift, nullptr, 1, nullptr, c1, 0, cur16, c1_16, 0, &packed_c1));
                if (!packed_c1) c1_16 = nullptr;      // conv2 packs its fp32 input itself
            }
            REID_TRY(conv_gemm(ctx, A_IM2COL, c1, n, Ho, Wo, k.c, k.conv2_w, k.c, 3, 3, 1, 1, 9 * k.c, nullptr, nullptr, 0,
                               k.bn2_scale, k.bn2_shift, k.ds ? nullptr : cur, k.ds ? 0 : 1, b.stats, y, 0, c1_16));
        } else {
        // conv1 (raw) + per-(image, channel) sum / sumsq partials for the InstanceNorm half
        REID_TRY(conv_gemm(ctx, A_IM2COL, cur, n, H, W, k.cin, k.conv1_w, k.c, 3, 3, k.stride, 1, 9 * k.cin, nullptr, nullptr, 0,
                           nullptr, nullptr, nullptr, 0, b.stats, c1));
        REID_TRY(launch_norm_finalize(ctx, b.stats, n, tiles, k.c, half, hw, k.in_gamma, k.in_beta, k.bn1_scale, k.bn1_shift,
                                      b.a_scale, b.a_shift));
        // conv2 with IBN/BN + ReLU fused into its loader; BN2 (+ identity residual + ReLU for blocks without
        // downsample: block_pre is the whole BasicBlock_IBN) and SE average-pool partials in its epilogue
        REID_TRY(conv_gemm(ctx, A_IM2COL, c1, n, Ho, Wo, k.c, k.conv2_w, k.c, 3, 3, 1, 1, 9 * k.c, b.a_scale, b.a_shift, 1,
                           k.bn2_scale, k.bn2_shift, k.ds ? nullptr : cur, k.ds ? 0 : 1, b.stats, y));
        }
        const float* shortcut = cur;
        if (k.ds) {
            REID_TRY(conv_gemm(ctx, A_IM2COL, cur, n, H, W, k.cin, k.ds_w, k.c, 1, 1, k.stride, 0, k.cin, nullptr, nullptr, 0,
                               k.ds_scale, k.ds_shift, nullptr, 0, nullptr, sc, 0, cur16));
            shortcut = sc;
        }
        float* out = c1;  // conv1 output is dead after conv2
        if (w.arch == 1) {          // CARes18_IBN: TripletAttention + shortcut + ReLU (CARes18.py:150-157)
            REID_TRY(launch_ta_tail(ctx, y, shortcut, n, Ho, Wo, k.c, k.ta, out));
        } else if (w.arch == 2) {   // EMARes18_IBN: EMA + shortcut + ReLU (EMA_Res18.py:79-86)
            REID_TRY(launch_ema_tail(ctx, y, shortcut, n, Ho, Wo, k.c, k.ema, out));
        } else if (ctx->f32_conv == 1) {   // SE gate + combine in one launch
            // precision 2: the next block reads this one's output as [oh | ol'] (conv1, shortcut conv) and, when it has no shortcut
            // conv, as the fp32 identity too; in front of a downsampling block whose two convolutions take the split path nobody
            // reads the fp32 form, so it is not written (debug-keep does: the stage taps)
            bool fp32_read = true;
            if (cur16 && !keep && i < 7 && w.blk[i + 1].ds && ctx->pack_epilogue) {
                const Se18Block& nx = w.blk[i + 1];
                fp32_read = !(conv_split_path(ctx, n, Ho, Wo, nx.cin, nx.c, 3, 3, nx.stride, 1) &&
                              conv_split_path(ctx, n, Ho, Wo, nx.cin, nx.c, 1, 1, nx.stride, 0));
            }
            REID_TRY(launch_se_tail(ctx, b.stats, n, tiles, k.c, k.mid, hw, k.se_w1, k.se_w2, y, shortcut, fp32_read ? out : nullptr,
                                    i < 7 ? cur16 : nullptr));
        } else {
            REID_TRY(launch_se_finalize(ctx, b.stats, n, tiles, k.c, k.mid, hw, k.se_w1, k.se_w2, b.se));
            REID_TRY(launch_se_combine(ctx, y, shortcut, b.se, n, hw, k.c, out));
        }
        b.stage[2 + i] = out;
        cur = out;
        H = Ho;
        W = Wo;
    }
    REID_TRY(launch_gem_neck(ctx, cur, n, H * W, 512, w.gem_p, w.neck_scale, w.neck_shift, b.gem, d_emb));
    {   // SERse18_IBN.forward(x, cam): + cam_factor * cam_bias[cam] on the BNNeck output, before the classifier (:269-271)
        const int32_t* d_cam;
        REID_TRY(ctx_take_side(ctx, n, w.num_cams, "reid_embed (camera bias)", &d_cam));
        if (d_cam) REID_TRY(launch_add_indexed_rows(ctx, d_emb, n, 1, 512, w.cam_bias, d_cam, w.cam_factor));
    }
    b.stage[10] = b.gem;
    if (d_logits) {
        if (!w.cls_w) {
            reid_set_error("logits requested but the weight blob has no classifier (cls.w)");
            return REID_ERR_STATE;
        }
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = d_emb; p.lda = 512;
        p.B = w.cls_w; p.ldb = 512;
        p.M = n; p.N = w.num_class; p.K = 512;
        p.C = d_logits; p.ldc = w.num_class;
        REID_TRY(launch_gemm_f32(ctx, A_DENSE, E_BIAS, p, REID_K_CONV_GEMM, 2.0 * n * w.num_class * 512,
                                 ((double)n * 512 + (double)w.num_class * 512 + (double)n * w.num_class) * 4.0));
    }
    ctx->last_n = n;
    ctx->last_f16 = false;
    for (int s = 0; s < 11; ++s) ctx->stage_ptr[s] = b.stage[s];
    return REID_OK;
}

// ------------------------------------------------------------------------------------------------ fp16 forward
int conv_gemm16(reid_ctx* ctx, int amode, const _Float16* x, int n, int H, int W, int Cin, const _Float16* wgt, int Cout,
                       int R, int S, int stride, int pad, int K, const float* col_scale, const float* col_shift,
                       const _Float16* residual, int relu, float* stats, _Float16* out, int Hp, int Wp) {
    Gemm16Params p;
    memset(&p, 0, sizeof(p));
    p.A = x;
    p.H = H; p.W = W; p.Cin = Cin; p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.Hp = Hp; p.Wp = Wp;
    p.Ho = (H + 2 * pad - R) / stride + 1;
    p.Wo = (W + 2 * pad - S) / stride + 1;
    p.B = wgt; p.ldb = amode == A16_STEM ? 256 : K;   // stem weights are stored [64][8 rows][32], 7 rows used
    p.M = n * p.Ho * p.Wo; p.N = Cout; p.K = K;
    p.C = out; p.ldc = Cout;
    p.col_scale = col_scale; p.col_shift = col_shift; p.residual = residual; p.relu = relu; p.stats = stats;
    p.zero_page = ctx->se18.zero_page;
    p.diag = ctx->conv_diag;
    const double ktrue = (double)R * S * (amode == A16_STEM ? 3 : Cin);
    const double flops = 2.0 * p.M * Cout * ktrue;
    const double bytes = ((double)n * H * W * (amode == A16_STEM ? 4 : Cin) + (double)p.M * Cout + (double)Cout * ktrue +
                          (residual ? (double)p.M * Cout : 0.0)) * 2.0;
    // 3x3 stride-1 convs: the LDS-halo kernel (8 compute + 4 loader waves) where it measured faster (tools/bench_conv_f16.py,
    // 256 crops): 32x16 maps 715 vs 604 TF, 16x8 x256ch 858 vs 745 TF, 16x8 x512ch 1015 vs 987 TF; the implicit GEMM keeps
    // Cout = 64 (489 vs 447 TF: nine K-tiles only, the halo kernel's longer prologue does not amortise)
    if (amode == A16_IM2COL && ctx->f16_halo && conv3x3_f16_supported(p)) {
        if (ctx->f16_halo == 2 || Cout >= 128) return launch_conv3x3_f16(ctx, p, REID_K_CONV_GEMM, flops, bytes);
    }
    return launch_gemm_f16(ctx, amode, p, REID_K_CONV_GEMM, flops, bytes);
}

static const int PAD_H = 262, PAD_W = 136;  // 256+6, 128+8: 3 zero rows/cols before, 3/5 after

// x: uint8 NHWC crops (is_u8) or fp32 NHWC, both [n][256][128][3] on the device
static int seres18_forward_f16(reid_ctx* ctx, const void* x, bool is_u8, int n, float* d_emb, float* d_logits) {
    Se18Weights& w = ctx->se18;
    if (!w.loaded) {
        reid_set_error("reid_embed_*: call reid_seres18_load first");
        return REID_ERR_STATE;
    }
    typedef _Float16 f16;
    const size_t per = 131072;
    const bool keep = ctx->debug_keep != 0;
    f16 *pad_in, *stem, *pool, *tbase;
    float *stats, *a_scale, *a_shift, *se, *gem;
    REID_TRY(ctx_ws(ctx, "se18h.pad", (size_t)n * PAD_H * PAD_W * 4 * 2, (void**)&pad_in));
    REID_TRY(ctx_ws(ctx, "se18h.stem", (size_t)n * 524288 * 2, (void**)&stem));
    REID_TRY(ctx_ws(ctx, "se18h.pool", (size_t)n * per * 2, (void**)&pool));
    REID_TRY(ctx_ws(ctx, "se18h.t", (size_t)n * per * 2 * (keep ? 4 * 8 : 4), (void**)&tbase));
    REID_TRY(ctx_ws(ctx, "se18.stats", (size_t)n * 2048 * 4, (void**)&stats));
    REID_TRY(ctx_ws(ctx, "se18.ascale", (size_t)n * 512 * 4, (void**)&a_scale));
    REID_TRY(ctx_ws(ctx, "se18.ashift", (size_t)n * 512 * 4, (void**)&a_shift));
    REID_TRY(ctx_ws(ctx, "se18.se", (size_t)n * 512 * 4, (void**)&se));
    REID_TRY(ctx_ws(ctx, "se18.gem", (size_t)n * 512 * 4, (void**)&gem));

    // The stem and layer-1 kernels give one whole image to a block: with fewer images than half the CUs (a tracking frame)
    // the tile-parallel GEMM kernels finish sooner (tools/bench_tracking.py: 758 vs 719 frames/s)
    const bool per_image_ok = n >= 128 || ctx->debug_keep == 2;
    const bool fused = ctx->f16_stem_fused && ctx->debug_keep != 1 && per_image_ok;   // debug_keep 1 keeps the unfused kernels (stage 0 = conv map)
    if (is_u8 && fused && ctx->f16_stem_fused == 2) {
        // uint8 crops straight into the fused stem: normalisation and zero padding happen while its LDS ring is filled
        REID_TRY(launch_stem_pool_f16(ctx, nullptr, (const uint8_t*)x, n, w.stem_w16s, w.stem_shift, pool));
    } else if (fused) {
        if (is_u8) REID_TRY(launch_prep_u8_pad_f16(ctx, (const uint8_t*)x, n, IMG_H, IMG_W, PAD_H, PAD_W, pad_in));
        else REID_TRY(launch_prep_f32_pad_f16(ctx, (const float*)x, n, IMG_H, IMG_W, PAD_H, PAD_W, pad_in));
        // conv 7x7 s2 + BN + MaxPool(3,2,1) in one kernel: the 1 MiB/crop conv map never reaches HBM (stem_pool_f16.hip)
        REID_TRY(launch_stem_pool_f16(ctx, pad_in, nullptr, n, w.stem_w16s, w.stem_shift, pool));
    } else {
        if (is_u8) REID_TRY(launch_prep_u8_pad_f16(ctx, (const uint8_t*)x, n, IMG_H, IMG_W, PAD_H, PAD_W, pad_in));
        else REID_TRY(launch_prep_f32_pad_f16(ctx, (const float*)x, n, IMG_H, IMG_W, PAD_H, PAD_W, pad_in));
        REID_TRY(conv_gemm16(ctx, A16_STEM, pad_in, n, IMG_H, IMG_W, 4, w.stem_w16, 64, 7, 7, 2, 3, 224, w.stem_scale,
                             w.stem_shift, nullptr, 0, nullptr, stem, PAD_H, PAD_W));
        REID_TRY(launch_maxpool3s2_f16(ctx, stem, n, 128, 64, 64, pool));
    }
    float* stage[11];
    stage[0] = (float*)stem;
    stage[1] = (float*)pool;

    const f16* cur = pool;
    int H = 64, W = 32;
    for (int i = 0; i < 8; ++i) {
        const Se18Block& k = w.blk[i];
        f16* tb[4];
        for (int j = 0; j < 4; ++j) tb[j] = tbase + ((size_t)(keep ? i * 4 : 0) + j) * n * per;
        f16* free_[3];
        int nf = 0;
        for (int j = 0; j < 4 && nf < 3; ++j)
            if (tb[j] != cur) free_[nf++] = tb[j];
        f16 *c1 = free_[0], *y = free_[1], *sc = free_[2];
        const int Ho = (H + 2 - 3) / k.stride + 1, Wo = (W + 2 - 3) / k.stride + 1;
        const int hw = Ho * Wo, tiles = hw / 128;
        const int half = k.ibn ? k.c / 2 : 0;
        // layer 1 (64 -> 64 on 64 x 32): register-resident-weight kernel, statistics per image (tiles = 1)
        const bool c64 = ctx->f16_c64 && per_image_ok && i < 2 && conv3x3_c64_f16_supported(H, W, k.cin, k.c, 3, 3, k.stride, 1) && !k.ds;
        if (c64) {
            REID_TRY(launch_conv3x3_c64_f16(ctx, cur, n, w.h(k.conv1_w), nullptr, nullptr, 0, stats, c1, w.zero_page));
            REID_TRY(launch_norm_apply_f16(ctx, c1, stats, n, 1, k.c, half, hw, k.in_gamma, k.in_beta, k.bn1_scale, k.bn1_shift));
        } else if (k.ibn) {
            // conv1 raw + per-(image, channel) statistics, then InstanceNorm/BN + ReLU in place
            REID_TRY(conv_gemm16(ctx, A16_IM2COL, cur, n, H, W, k.cin, w.h(k.conv1_w), k.c, 3, 3, k.stride, 1, 9 * k.cin, nullptr,
                                 nullptr, nullptr, 0, stats, c1));
            if (ctx->debug_keep == 1) {   // unfused reference sequence
                REID_TRY(launch_norm_finalize(ctx, stats, n, tiles, k.c, half, hw, k.in_gamma, k.in_beta, k.bn1_scale, k.bn1_shift,
                                              a_scale, a_shift));
                REID_TRY(launch_affine_relu_f16(ctx, c1, a_scale, a_shift, n, hw, k.c));
            } else {
                REID_TRY(launch_norm_apply_f16(ctx, c1, stats, n, tiles, k.c, half, hw, k.in_gamma, k.in_beta, k.bn1_scale, k.bn1_shift));
            }
        } else {
            // plain BatchNorm (layer 4): BN + ReLU go straight into the conv1 epilogue
            REID_TRY(conv_gemm16(ctx, A16_IM2COL, cur, n, H, W, k.cin, w.h(k.conv1_w), k.c, 3, 3, k.stride, 1, 9 * k.cin,
                                 k.bn1_scale, k.bn1_shift, nullptr, 1, nullptr, c1));
        }
        if (c64 && ctx->f16_c64 == 2) {
            // conv2 + residual + ReLU (Q5) + SE gate + combine in one kernel: `out` is the block output, y never reaches HBM
            f16* outb = y;   // c1 is this launch's input, cur its shortcut: the block output goes to the third buffer
            REID_TRY(launch_conv3x3_c64_f16(ctx, c1, n, w.l1_conv2_w16s[i], k.bn2_shift, cur, 1, nullptr, outb, w.zero_page, k.se_w1,
                                            k.se_w2));
            stage[2 + i] = (float*)outb;
            cur = outb;
            H = Ho;
            W = Wo;
            continue;
        } else if (c64) {
            REID_TRY(launch_conv3x3_c64_f16(ctx, c1, n, w.l1_conv2_w16s[i], k.bn2_shift, cur, 1, stats, y, w.zero_page));   // residual + ReLU: Q5
        } else {
            REID_TRY(conv_gemm16(ctx, A16_IM2COL, c1, n, Ho, Wo, k.c, w.h(k.conv2_w), k.c, 3, 3, 1, 1, 9 * k.c, k.bn2_scale,
                                 k.bn2_shift, k.ds ? nullptr : cur, k.ds ? 0 : 1, stats, y));
        }
        const f16* shortcut = cur;
        if (k.ds) {
            REID_TRY(conv_gemm16(ctx, A16_IM2COL, cur, n, H, W, k.cin, w.h(k.ds_w), k.c, 1, 1, k.stride, 0, k.cin, k.ds_scale,
                                 k.ds_shift, nullptr, 0, nullptr, sc));
            shortcut = sc;
        }
        f16* out = c1;
        if (ctx->f16_se_tail && ctx->debug_keep != 1) {   // gate + combine in one launch, sliced per image when there are few images
            REID_TRY(launch_se_tail_f16(ctx, stats, n, c64 ? 1 : tiles, k.c, k.mid, hw, k.se_w1, k.se_w2, y, shortcut, out));
        } else {
            REID_TRY(launch_se_finalize(ctx, stats, n, c64 ? 1 : tiles, k.c, k.mid, hw, k.se_w1, k.se_w2, se));
            REID_TRY(launch_se_combine_f16(ctx, y, shortcut, se, n, hw, k.c, out));
        }
        stage[2 + i] = (float*)out;
        cur = out;
        H = Ho;
        W = Wo;
    }
    REID_TRY(launch_gem_neck_f16(ctx, cur, n, H * W, 512, w.gem_p, w.neck_scale, w.neck_shift, gem, d_emb));
    {
        const int32_t* d_cam;
        REID_TRY(ctx_take_side(ctx, n, w.num_cams, "reid_embed (camera bias)", &d_cam));
        if (d_cam) REID_TRY(launch_add_indexed_rows(ctx, d_emb, n, 1, 512, w.cam_bias, d_cam, w.cam_factor));
    }
    stage[10] = gem;
    if (d_logits) {
        if (!w.cls_w) {
            reid_set_error("logits requested but the weight blob has no classifier (cls.w)");
            return REID_ERR_STATE;
        }
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = d_emb; p.lda = 512;
        p.B = w.cls_w; p.ldb = 512;
        p.M = n; p.N = w.num_class; p.K = 512;
        p.C = d_logits; p.ldc = w.num_class;
        REID_TRY(launch_gemm_f32(ctx, A_DENSE, E_BIAS, p, REID_K_CONV_GEMM, 2.0 * n * w.num_class * 512,
                                 ((double)n * 512 + (double)w.num_class * 512 + (double)n * w.num_class) * 4.0));
    }
    ctx->last_n = n;
    ctx->last_f16 = true;
    for (int s2 = 0; s2 < 11; ++s2) ctx->stage_ptr[s2] = stage[s2];
    return REID_OK;
}

// (Replaying a captured hipGraph for small batches was tried: the host saves ~150 us of launch calls per tracking frame, but
// the device-side replay is slower than plain launches and the frame pipeline already hides the host - 1421 vs 1516 frames/s.)
static int seres18_run(reid_ctx* ctx, const void* x, bool is_u8, int n, float* d_emb, float* d_logits) {
    // the sibling backbones (CARes18 / EMARes18) exist in the reference's arithmetic only
    return (ctx->precision == 1 && ctx->se18.arch == 0) ? seres18_forward_f16(ctx, x, is_u8, n, d_emb, d_logits)
                                                         : seres18_forward(ctx, x, is_u8, n, d_emb, d_logits);
}

static const size_t kStageElems[11] = {524288, 131072, 131072, 131072, 65536, 65536, 32768, 32768, 65536, 65536, 512};

extern "C" int reid_debug_stage(reid_ctx* ctx, int stage, float* out, size_t max_floats, size_t* count) {
    ARG_CHECK(ctx && stage >= 0 && stage < 11 && out);
    CTX_GUARD(ctx);
    if (!ctx->debug_keep || ctx->last_n <= 0) {
        reid_set_error("reid_debug_stage: enable reid_ctx_set_debug_keep before the embed call");
        return REID_ERR_STATE;
    }
    const size_t total = kStageElems[stage] * (size_t)ctx->last_n;
    if (count) *count = total;
    const size_t ncopy = total < max_floats ? total : max_floats;
    if (ctx->last_f16 && stage < 10) {   // fp16 path: activations are stored as f16, widen on the host
        std::vector<_Float16> tmp(ncopy);
        HIP_TRY(hipMemcpyAsync(tmp.data(), ctx->stage_ptr[stage], ncopy * 2, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < ncopy; ++i) out[i] = (float)tmp[i];
        return REID_OK;
    }
    HIP_TRY(hipMemcpyAsync(out, ctx->stage_ptr[stage], ncopy * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}

int ctx_pipe_events(reid_ctx* ctx, int passes) {
    if (!ctx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    while ((int)ctx->pipe_ev.size() < 2 * passes) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->pipe_ev.push_back(e);
    }
    return REID_OK;
}

extern "C" int reid_embed_u8_dev(reid_ctx* ctx, const uint8_t* d_crops, int n, float* d_emb, float* d_logits) {
    ARG_CHECK(ctx && d_crops && d_emb && n >= 0);
    CTX_ENTER(ctx);
    const int nc = ctx->se18.num_class;
    for (int i = 0; i < n; i += ctx->chunk) {
        const int m = n - i < ctx->chunk ? n - i : ctx->chunk;
        REID_TRY(seres18_run(ctx, d_crops + (size_t)i * IMG_H * IMG_W * 3, true, m, d_emb + (size_t)i * 512,
                                 d_logits ? d_logits + (size_t)i * nc : nullptr));
    }
    return REID_OK;
}

extern "C" int reid_embed_u8(reid_ctx* ctx, const uint8_t* crops, int n, float* emb, float* logits) {
    ARG_CHECK(ctx && crops && emb && n >= 0);
    CTX_ENTER(ctx);
    if (n == 0) return REID_OK;
    const int nc = ctx->se18.num_class;
    uint8_t* d_in;
    float *d_emb, *d_log = nullptr;
    const size_t crop_b = (size_t)IMG_H * IMG_W * 3;
    REID_TRY(ctx_ws(ctx, "io.in", (size_t)n * crop_b, (void**)&d_in));
    REID_TRY(ctx_ws(ctx, "io.emb", (size_t)n * 512 * 4, (void**)&d_emb));
    if (logits) REID_TRY(ctx_ws(ctx, "io.logits", (size_t)n * nc * 4 + 16, (void**)&d_log));
    REID_TRY(host_passes(
        ctx, n, ctx->chunk,
        [&](int i, int m, hipStream_t s) -> int {
            HIP_TRY(hipMemcpyAsync(d_in + (size_t)i * crop_b, crops + (size_t)i * crop_b, (size_t)m * crop_b, hipMemcpyHostToDevice, s));
            return REID_OK;
        },
        [&](int i, int m) -> int {
            return seres18_run(ctx, d_in + (size_t)i * crop_b, true, m, d_emb + (size_t)i * 512, d_log ? d_log + (size_t)i * nc : nullptr);
        },
        [&](int i, int m, hipStream_t s) -> int {
            HIP_TRY(hipMemcpyAsync(emb + (size_t)i * 512, d_emb + (size_t)i * 512, (size_t)m * 512 * 4, hipMemcpyDeviceToHost, s));
            if (logits) HIP_TRY(hipMemcpyAsync(logits + (size_t)i * nc, d_log + (size_t)i * nc, (size_t)m * nc * 4, hipMemcpyDeviceToHost, s));
            return REID_OK;
        }));
    return ctx_fault_status(ctx);
}

extern "C" int reid_embed_f32_nchw_dev(reid_ctx* ctx, const float* d_x, int n, float* d_emb, float* d_logits) {
    ARG_CHECK(ctx && d_x && d_emb && n >= 0);
    CTX_ENTER(ctx);
    const int nc = ctx->se18.num_class;
    const size_t img = (size_t)IMG_H * IMG_W * 3;
    for (int i = 0; i < n; i += ctx->chunk) {
        const int m = n - i < ctx->chunk ? n - i : ctx->chunk;
        float* nhwc;
        REID_TRY(ctx_ws(ctx, "se18.in_nhwc", (size_t)m * img * 4, (void**)&nhwc));
        REID_TRY(launch_nchw_to_nhwc3(ctx, d_x + (size_t)i * img, m, IMG_H, IMG_W, nhwc));
        REID_TRY(seres18_run(ctx, nhwc, false, m, d_emb + (size_t)i * 512, d_logits ? d_logits + (size_t)i * nc : nullptr));
    }
    return REID_OK;
}

extern "C" int reid_embed_f32_nchw(reid_ctx* ctx, const float* x, int n, float* emb, float* logits) {
    ARG_CHECK(ctx && x && emb && n >= 0);
    CTX_ENTER(ctx);
    if (n == 0) return REID_OK;
    const int nc = ctx->se18.num_class;
    const size_t img = (size_t)IMG_H * IMG_W * 3;
    float *d_in, *d_emb, *d_log = nullptr;
    REID_TRY(ctx_ws(ctx, "io.in", (size_t)n * img * 4, (void**)&d_in));
    REID_TRY(ctx_ws(ctx, "io.emb", (size_t)n * 512 * 4, (void**)&d_emb));
    if (logits) REID_TRY(ctx_ws(ctx, "io.logits", (size_t)n * nc * 4 + 16, (void**)&d_log));
    REID_TRY(host_passes(
        ctx, n, ctx->chunk,
        [&](int i, int m, hipStream_t s) -> int {
            HIP_TRY(hipMemcpyAsync(d_in + (size_t)i * img, x + (size_t)i * img, (size_t)m * img * 4, hipMemcpyHostToDevice, s));
            return REID_OK;
        },
        [&](int i, int m) -> int {
            float* nhwc;
            REID_TRY(ctx_ws(ctx, "se18.in_nhwc", (size_t)m * img * 4, (void**)&nhwc));
            REID_TRY(launch_nchw_to_nhwc3(ctx, d_in + (size_t)i * img, m, IMG_H, IMG_W, nhwc));
            return seres18_run(ctx, nhwc, false, m, d_emb + (size_t)i * 512, d_log ? d_log + (size_t)i * nc : nullptr);
        },
        [&](int i, int m, hipStream_t s) -> int {
            HIP_TRY(hipMemcpyAsync(emb + (size_t)i * 512, d_emb + (size_t)i * 512, (size_t)m * 512 * 4, hipMemcpyDeviceToHost, s));
            if (logits) HIP_TRY(hipMemcpyAsync(logits + (size_t)i * nc, d_log + (size_t)i * nc, (size_t)m * nc * 4, hipMemcpyDeviceToHost, s));
            return REID_OK;
        }));
    return ctx_fault_status(ctx);
}

// Enqueue only (no synchronisation): upload of the ragged crops, device-side resize + normalise, forward.  `tag` names the
// device buffers (the frame pipeline keeps one set per frame slot); offsets / hw must be pinned or outlive the stream's work.
int embed_ragged_enqueue(reid_ctx* ctx, const char* tag, const uint8_t* packed, const int64_t* offsets, const int32_t* hw, int n,
                         float** d_emb_out, float** d_log_out, bool side_copy) {
    const int nc = ctx->se18.num_class;
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        ARG_CHECK(hw[2 * i] >= 1 && hw[2 * i + 1] >= 1 && offsets[i] >= 0);
        const size_t end = (size_t)offsets[i] + (size_t)hw[2 * i] * hw[2 * i + 1] * 3;
        if (end > total) total = end;
    }
    uint8_t* d_pk;
    char* d_meta;
    float *d_emb, *d_log = nullptr;
    const std::string t(tag);
    REID_TRY(ctx_ws(ctx, (t + ".in").c_str(), total, (void**)&d_pk));
    REID_TRY(ctx_ws(ctx, (t + ".meta").c_str(), (size_t)n * 16, (void**)&d_meta));
    long long* d_off = (long long*)d_meta;
    int* d_hw = (int*)(d_meta + (size_t)n * 8);
    REID_TRY(ctx_ws(ctx, (t + ".emb").c_str(), (size_t)(n + 1) * 512 * 4, (void**)&d_emb));   // + 1: the padding row of reid_frame_gather
    if (d_log_out) REID_TRY(ctx_ws(ctx, (t + ".logits").c_str(), (size_t)n * nc * 4 + 16, (void**)&d_log));
    // side_copy (frame pipeline, pinned sources): the upload runs on a copy stream beside the previous frame's kernels
    hipStream_t cs = ctx->stream;
    if (side_copy) {
        if (!ctx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        if (!ctx->copy_ev) HIP_TRY(hipEventCreateWithFlags(&ctx->copy_ev, hipEventDisableTiming));
        cs = ctx->copy_stream;
    }
    HIP_TRY(hipMemcpyAsync(d_pk, packed, total, hipMemcpyHostToDevice, cs));
    if ((const char*)hw == (const char*)offsets + (size_t)n * 8) {
        HIP_TRY(hipMemcpyAsync(d_meta, offsets, (size_t)n * 16, hipMemcpyHostToDevice, cs));
    } else {
        HIP_TRY(hipMemcpyAsync(d_off, offsets, (size_t)n * 8, hipMemcpyHostToDevice, cs));
        HIP_TRY(hipMemcpyAsync(d_hw, hw, (size_t)n * 8, hipMemcpyHostToDevice, cs));
    }
    if (side_copy) {
        HIP_TRY(hipEventRecord(ctx->copy_ev, cs));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->copy_ev, 0));
    }
    const size_t img = (size_t)IMG_H * IMG_W * 3;
    for (int i = 0; i < n; i += ctx->chunk) {
        const int m = n - i < ctx->chunk ? n - i : ctx->chunk;
        float* nhwc;
        REID_TRY(ctx_ws(ctx, "se18.in_nhwc", (size_t)m * img * 4, (void**)&nhwc));
        REID_TRY(launch_resize_norm(ctx, d_pk, d_off + i, d_hw + 2 * i, m, IMG_H, IMG_W, 0, nhwc));
        REID_TRY(seres18_run(ctx, nhwc, false, m, d_emb + (size_t)i * 512, d_log ? d_log + (size_t)i * nc : nullptr));
    }
    *d_emb_out = d_emb;
    if (d_log_out) *d_log_out = d_log;
    return REID_OK;
}

extern "C" int reid_embed_ragged_u8(reid_ctx* ctx, const uint8_t* packed, const int64_t* offsets, const int32_t* hw, int n,
                                    float* emb, float* logits) {
    ARG_CHECK(ctx && packed && offsets && hw && emb && n >= 0);
    CTX_ENTER(ctx);
    if (n == 0) return REID_OK;
    const int nc = ctx->se18.num_class;
    float *d_emb, *d_log = nullptr;
    if (n <= ctx->chunk || !ctx->host_pipeline) {
        REID_TRY(embed_ragged_enqueue(ctx, "io", packed, offsets, hw, n, &d_emb, logits ? &d_log : nullptr, false));
        HIP_TRY(hipMemcpyAsync(emb, d_emb, (size_t)n * 512 * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (logits) HIP_TRY(hipMemcpyAsync(logits, d_log, (size_t)n * nc * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return ctx_fault_status(ctx);
    }
    // several passes: the bytes a pass reads are the span [lowest offset, highest end) of its crops (the usual packing - crop after
    // crop - makes the spans a partition of the buffer; any other layout copies some bytes twice, which is harmless)
    const int passes = (n + ctx->chunk - 1) / ctx->chunk;
    std::vector<size_t> lo(passes, (size_t)-1), hi(passes, 0);
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        ARG_CHECK(hw[2 * i] >= 1 && hw[2 * i + 1] >= 1 && offsets[i] >= 0);
        const size_t b = (size_t)offsets[i], e = b + (size_t)hw[2 * i] * hw[2 * i + 1] * 3;
        const int k = i / ctx->chunk;
        if (b < lo[k]) lo[k] = b;
        if (e > hi[k]) hi[k] = e;
        if (e > total) total = e;
    }
    uint8_t* d_pk;
    char* d_meta;
    REID_TRY(ctx_ws(ctx, "io.in", total, (void**)&d_pk));
    REID_TRY(ctx_ws(ctx, "io.meta", (size_t)n * 16, (void**)&d_meta));
    long long* d_off = (long long*)d_meta;
    int* d_hw = (int*)(d_meta + (size_t)n * 8);
    REID_TRY(ctx_ws(ctx, "io.emb", (size_t)(n + 1) * 512 * 4, (void**)&d_emb));
    if (logits) REID_TRY(ctx_ws(ctx, "io.logits", (size_t)n * nc * 4 + 16, (void**)&d_log));
    const size_t img = (size_t)IMG_H * IMG_W * 3;
    REID_TRY(host_passes(
        ctx, n, ctx->chunk,
        [&](int i, int m, hipStream_t s) -> int {
            const int k = i / ctx->chunk;
            if (i == 0) {
                HIP_TRY(hipMemcpyAsync(d_off, offsets, (size_t)n * 8, hipMemcpyHostToDevice, s));
                HIP_TRY(hipMemcpyAsync(d_hw, hw, (size_t)n * 8, hipMemcpyHostToDevice, s));
            }
            HIP_TRY(hipMemcpyAsync(d_pk + lo[k], packed + lo[k], hi[k] - lo[k], hipMemcpyHostToDevice, s));
            return REID_OK;
        },
        [&](int i, int m) -> int {
            float* nhwc;
            REID_TRY(ctx_ws(ctx, "se18.in_nhwc", (size_t)m * img * 4, (void**)&nhwc));
            REID_TRY(launch_resize_norm(ctx, d_pk, d_off + i, d_hw + 2 * i, m, IMG_H, IMG_W, 0, nhwc));
            return seres18_run(ctx, nhwc, false, m, d_emb + (size_t)i * 512, d_log ? d_log + (size_t)i * nc : nullptr);
        },
        [&](int i, int m, hipStream_t s) -> int {
            HIP_TRY(hipMemcpyAsync(emb + (size_t)i * 512, d_emb + (size_t)i * 512, (size_t)m * 512 * 4, hipMemcpyDeviceToHost, s));
            if (logits) HIP_TRY(hipMemcpyAsync(logits + (size_t)i * nc, d_log + (size_t)i * nc, (size_t)m * nc * 4, hipMemcpyDeviceToHost, s));
            return REID_OK;
        }));
    return ctx_fault_status(ctx);
}

// Crops given as windows of ONE frame: DeepSort._get_features ([external] deep_sort.py) slices ori_img[y1:y2, x1:x2] per box
// on the host and Extractor._preprocess (feature_extractor.py:31-46) resizes each slice with cv2; here the frame is uploaded
// once and every window is resized straight out of it on the device (same bilinear taps, clamped to the WINDOW's border).
extern "C" int reid_embed_frame_u8(reid_ctx* ctx, const uint8_t* frame, int fh, int fw, const int32_t* boxes_xyxy, int n,
                                   float* emb, float* logits) {
    ARG_CHECK(ctx && frame && boxes_xyxy && emb && fh >= 1 && fw >= 1 && n >= 0);
    CTX_ENTER(ctx);
    if (n == 0) return REID_OK;
    const int nc = ctx->se18.num_class;
    std::vector<long long> off(n);
    std::vector<int> hw(2 * n);
    for (int i = 0; i < n; ++i) {
        const int x1 = boxes_xyxy[4 * i], y1 = boxes_xyxy[4 * i + 1], x2 = boxes_xyxy[4 * i + 2], y2 = boxes_xyxy[4 * i + 3];
        ARG_CHECK(x1 >= 0 && y1 >= 0 && x2 <= fw && y2 <= fh && x2 > x1 && y2 > y1);   // an empty slice fails in cv2.resize too
        off[i] = ((long long)y1 * fw + x1) * 3;
        hw[2 * i] = y2 - y1;
        hw[2 * i + 1] = x2 - x1;
    }
    uint8_t* d_fr;
    long long* d_off;
    int* d_hw;
    float *d_emb, *d_log = nullptr;
    const size_t total = (size_t)fh * fw * 3;
    REID_TRY(ctx_ws(ctx, "io.in", total, (void**)&d_fr));
    REID_TRY(ctx_ws(ctx, "io.off", (size_t)n * 8, (void**)&d_off));
    REID_TRY(ctx_ws(ctx, "io.hw", (size_t)n * 8, (void**)&d_hw));
    REID_TRY(ctx_ws(ctx, "io.emb", (size_t)n * 512 * 4, (void**)&d_emb));
    if (logits) REID_TRY(ctx_ws(ctx, "io.logits", (size_t)n * nc * 4 + 16, (void**)&d_log));
    HIP_TRY(hipMemcpyAsync(d_fr, frame, total, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_off, off.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_hw, hw.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // off / hw are locals
    const size_t img = (size_t)IMG_H * IMG_W * 3;
    for (int i = 0; i < n; i += ctx->chunk) {
        const int m = n - i < ctx->chunk ? n - i : ctx->chunk;
        float* nhwc;
        REID_TRY(ctx_ws(ctx, "se18.in_nhwc", (size_t)m * img * 4, (void**)&nhwc));
        REID_TRY(launch_resize_norm(ctx, d_fr, d_off + i, d_hw + 2 * i, m, IMG_H, IMG_W, fw, nhwc));
        REID_TRY(seres18_run(ctx, nhwc, false, m, d_emb + (size_t)i * 512, d_log ? d_log + (size_t)i * nc : nullptr));
    }
    HIP_TRY(hipMemcpyAsync(emb, d_emb, (size_t)n * 512 * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (logits) HIP_TRY(hipMemcpyAsync(logits, d_log, (size_t)n * nc * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ctx_fault_status(ctx);
}

// ------------------------------------------------------------------------------------------------ matching
// pads d to a multiple of 4 when needed (zero columns change neither dot products nor norms)
static int pad_rows(reid_ctx* ctx, const char* name, const float* d_x, int m, int d, const float** out, int* ld) {
    if (d % 4 == 0 && ((uintptr_t)d_x % 16) == 0) {
        *out = d_x;
        *ld = d;
        return REID_OK;
    }
    const int dp = (d + 3) / 4 * 4;
    float* buf;
    REID_TRY(ctx_ws(ctx, name, (size_t)m * dp * 4, (void**)&buf));
    HIP_TRY(hipMemsetAsync(buf, 0, (size_t)m * dp * 4, ctx->stream));
    HIP_TRY(hipMemcpy2DAsync(buf, (size_t)dp * 4, d_x, (size_t)d * 4, (size_t)d * 4, m, hipMemcpyDeviceToDevice, ctx->stream));
    *out = buf;
    *ld = dp;
    return REID_OK;
}

extern "C" int reid_distmat_dev(reid_ctx* ctx, const float* d_x, int m, const float* d_y, int n, int d, int metric,
                                float* d_out) {
    ARG_CHECK(ctx && d_x && d_y && d_out && m >= 0 && n >= 0 && d >= 1);
    CTX_GUARD(ctx);
    ARG_CHECK(metric >= REID_METRIC_L2 && metric <= REID_METRIC_DOT);
    if (m == 0 || n == 0) return REID_OK;
    const float *xp, *yp;
    int ldx, ldy;
    REID_TRY(pad_rows(ctx, "dist.xpad", d_x, m, d, &xp, &ldx));
    REID_TRY(pad_rows(ctx, "dist.ypad", d_y, n, d, &yp, &ldy));
    float *xx = nullptr, *yy = nullptr;
    if (metric != REID_METRIC_DOT) {
        REID_TRY(ctx_ws(ctx, "dist.xx", (size_t)m * 4, (void**)&xx));
        REID_TRY(ctx_ws(ctx, "dist.yy", (size_t)n * 4, (void**)&yy));
        REID_TRY(launch_row_sqnorm(ctx, xp, m, ldx, ldx, xx));
        REID_TRY(launch_row_sqnorm(ctx, yp, n, ldy, ldy, yy));
    }
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = xp; p.lda = ldx;
    p.B = yp; p.ldb = ldy;
    p.M = m; p.N = n; p.K = ldx;
    p.C = d_out; p.ldc = n;
    p.row_sq = xx; p.col_sq = yy; p.metric = metric;
    return launch_gemm_f32(ctx, A_DENSE, E_DIST, p, REID_K_DIST_GEMM, 2.0 * m * n * d,
                           4.0 * ((double)m * d + (double)n * d + (double)m * n));
}

struct HostIO {
    reid_ctx* ctx;
    float *dx = nullptr, *dy = nullptr;
    int upload(const float* x, int m, const float* y, int n, int d) {
        REID_TRY(ctx_ws(ctx, "io.x", (size_t)(m ? m : 1) * d * 4, (void**)&dx));
        REID_TRY(ctx_ws(ctx, "io.y", (size_t)(n ? n : 1) * d * 4, (void**)&dy));
        if (m) HIP_TRY(hipMemcpyAsync(dx, x, (size_t)m * d * 4, hipMemcpyHostToDevice, ctx->stream));
        if (n) HIP_TRY(hipMemcpyAsync(dy, y, (size_t)n * d * 4, hipMemcpyHostToDevice, ctx->stream));
        return REID_OK;
    }
};

extern "C" int reid_distmat(reid_ctx* ctx, const float* x, int m, const float* y, int n, int d, int metric, float* out) {
    ARG_CHECK(ctx && x && y && out && m >= 0 && n >= 0 && d >= 1);
    CTX_GUARD(ctx);
    if (m == 0 || n == 0) return REID_OK;
    HostIO io{ctx};
    REID_TRY(io.upload(x, m, y, n, d));
    float* d_out;
    REID_TRY(ctx_ws(ctx, "io.dist", (size_t)m * n * 4, (void**)&d_out));
    REID_TRY(reid_distmat_dev(ctx, io.dx, m, io.dy, n, d, metric, d_out));
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)m * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}

// zero-padded copy with rows of a multiple of `mult` floats (mult = 4: 16-byte rows; 32: whole K-tiles of the LDS-DMA loops)
static int pad_rows_to(reid_ctx* ctx, const char* name, const float* d_x, int m, int d, int mult, const float** out, int* ld) {
    if (d % mult == 0 && ((uintptr_t)d_x % 16) == 0) {
        *out = d_x;
        *ld = d;
        return REID_OK;
    }
    const int dp = (d + mult - 1) / mult * mult;
    float* buf;
    REID_TRY(ctx_ws(ctx, name, (size_t)m * dp * 4, (void**)&buf));
    HIP_TRY(hipMemsetAsync(buf, 0, (size_t)m * dp * 4, ctx->stream));
    HIP_TRY(hipMemcpy2DAsync(buf, (size_t)dp * 4, d_x, (size_t)d * 4, (size_t)d * 4, m, hipMemcpyDeviceToDevice, ctx->stream));
    *out = buf;
    *ld = dp;
    return REID_OK;
}

// The k smallest distances of every row of x to the rows of y with the selection fused into the distance GEMM
// (dist_select.hip): no m x n matrix.  Returns 1 when the shape is outside the fused kernel's range (k > 64, huge row pitch):
// the caller then takes the two-pass path.
static int select_fused_dev(reid_ctx* ctx, const float* d_x, int m, const float* d_y, int n, int d, int metric, int k, float* d_D,
                            int32_t* d_I) {
    // small problems keep the two-pass form: their matrix is a few MB, and without a sample bound the sweep's first tile would
    // append every element
    constexpr int sample = 1024;
    if (k > SEL_KMAX || n < 2 * sample || (long long)m * n < (1ll << 18)) return 1;
    const float *xp, *yp;
    int ldx, ldy;
    REID_TRY(pad_rows_to(ctx, "sel.xpad", d_x, m, d, 64, &xp, &ldx));   // whole pairs of K-tiles
    REID_TRY(pad_rows_to(ctx, "sel.ypad", d_y, n, d, 64, &yp, &ldy));
    SelectParams p;
    memset(&p, 0, sizeof(p));
    p.A = xp; p.lda = ldx;
    p.B = yp; p.ldb = ldy;
    p.M = m; p.N = n; p.K = ldx;
    p.metric = metric; p.k = k;
    p.exp_skip = ctx->select_exp;   // 0 in the product; experiments set it through libreid_hip_debug.so (reid_debug_select_exp)
    if (!dist_select_supported(p)) return 1;
    if (metric != REID_METRIC_DOT) {
        float *xx, *yy;
        REID_TRY(ctx_ws(ctx, "dist.xx", (size_t)m * 4, (void**)&xx));
        REID_TRY(ctx_ws(ctx, "dist.yy", (size_t)n * 4, (void**)&yy));
        REID_TRY(launch_row_sqnorm(ctx, xp, m, ldx, ldx, xx));
        REID_TRY(launch_row_sqnorm(ctx, yp, n, ldy, ldy, yy));
        p.row_sq = xx; p.col_sq = yy;
    }
    p.S = select_segments(m, n);
    if (k == 1) {
        REID_TRY(ctx_ws(ctx, "sel.final", (size_t)m * p.S * 8, (void**)&p.final_keys));
    } else {
        REID_TRY(ctx_ws(ctx, "sel.lists", (size_t)m * p.S * SEL_CAP * 8, (void**)&p.lists));
        REID_TRY(ctx_ws(ctx, "sel.counts", (size_t)m * p.S * 4, (void**)&p.counts));
        // per-row bound from a SAMPLE of y (its first 1024 rows - part of y, so a bound of the sample's k-th smallest holds for all
        // of y); the arg-min form needs none (its threshold is the running minimum itself)
        REID_TRY(ctx_ws(ctx, "sel.gmin", (size_t)m * k * 4, (void**)&p.gmin));
        HIP_TRY(hipMemsetAsync(p.gmin, 0xff, (size_t)m * k * 4, ctx->stream));
        SelectParams ps = p;
        ps.N = sample;
        REID_TRY(launch_dist_bound(ctx, ps));
    }
    REID_TRY(launch_dist_select(ctx, p, d_D, d_I));
    if (p.exp_skip == 4 && k > 1) {   // diagnostics: how long are the candidate lists when the sweep ends?
        std::vector<int> h((size_t)m * p.S);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpy(h.data(), p.counts, h.size() * 4, hipMemcpyDeviceToHost));
        long long sum = 0;
        int mx = 0;
        for (int v : h) { sum += v; mx = v > mx ? v : mx; }
        fprintf(stderr, "[dist_select] m %d n %d k %d S %d: candidates per (row, segment) mean %.1f max %d, per row %.0f\n", m, n, k, p.S,
                (double)sum / h.size(), mx, (double)sum / m);
    }
    return REID_OK;
}

extern "C" int reid_argmin_rows_dev(reid_ctx* ctx, const float* d_x, int m, const float* d_y, int n, int d, int metric,
                                    int32_t* d_idx, float* d_val) {
    ARG_CHECK(ctx && d_x && d_y && d_idx && m >= 0 && n >= 1 && d >= 1);
    CTX_GUARD(ctx);
    ARG_CHECK(metric >= REID_METRIC_L2 && metric <= REID_METRIC_DOT);
    if (m == 0) return REID_OK;
    const int st = ctx->select_two_pass ? 1 : select_fused_dev(ctx, d_x, m, d_y, n, d, metric, 1, d_val, d_idx);
    if (st != 1) return st;
    float* dist;
    REID_TRY(ctx_ws(ctx, "sel.dist", (size_t)m * n * 4, (void**)&dist));
    REID_TRY(reid_distmat_dev(ctx, d_x, m, d_y, n, d, metric, dist));
    return launch_argmin_rows(ctx, dist, m, n, n, d_idx, d_val);
}

extern "C" int reid_argmin_rows(reid_ctx* ctx, const float* x, int m, const float* y, int n, int d, int metric, int32_t* idx,
                                float* val) {
    ARG_CHECK(ctx && x && y && idx && m >= 0 && n >= 1 && d >= 1);
    CTX_GUARD(ctx);
    if (m == 0) return REID_OK;
    HostIO io{ctx};
    REID_TRY(io.upload(x, m, y, n, d));
    int32_t* d_idx;
    float* d_val;
    REID_TRY(ctx_ws(ctx, "io.idx", (size_t)m * 4, (void**)&d_idx));
    REID_TRY(ctx_ws(ctx, "io.val", (size_t)m * 4, (void**)&d_val));
    REID_TRY(reid_argmin_rows_dev(ctx, io.dx, m, io.dy, n, d, metric, d_idx, d_val));
    HIP_TRY(hipMemcpyAsync(idx, d_idx, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (val) HIP_TRY(hipMemcpyAsync(val, d_val, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}

extern "C" int reid_knn_dev(reid_ctx* ctx, const float* d_xq, int nq, const float* d_xb, int nb, int d, int k, float* d_D,
                            int32_t* d_I) {
    ARG_CHECK(ctx && d_D && d_I && nq >= 0 && nb >= 1 && k >= 1);
    CTX_GUARD(ctx);
    if (nq == 0) return REID_OK;
    if (!ctx->select_two_pass && knn_wide_eligible(ctx, nq, nb, d, k)) {
        // large searches (the re-ranking's 19 281 x 19 281 x 1 263): candidates on the f16 matrix pipe, exact fp32 refinement - the
        // numbers and the order of the fused fp32 search below (knn_wide.hip)
        const float *xp, *yp;
        int ldx, ldy;
        REID_TRY(pad_rows_to(ctx, "sel.xpad", d_xq, nq, d, 64, &xp, &ldx));
        REID_TRY(pad_rows_to(ctx, "sel.ypad", d_xb, nb, d, 64, &yp, &ldy));
        float *xx, *yy;
        REID_TRY(ctx_ws(ctx, "dist.xx", (size_t)nq * 4, (void**)&xx));
        REID_TRY(ctx_ws(ctx, "dist.yy", (size_t)nb * 4, (void**)&yy));
        REID_TRY(launch_row_sqnorm(ctx, xp, nq, ldx, ldx, xx));
        REID_TRY(launch_row_sqnorm(ctx, yp, nb, ldy, ldy, yy));
        return knn_wide_dev(ctx, xp, nq, yp, nb, ldx, xx, yy, k, d_D, d_I);
    }
    if (!ctx->select_two_pass) {   // fused distance + selection: the nq x nb matrix is never written
        const int st = select_fused_dev(ctx, d_xq, nq, d_xb, nb, d, REID_METRIC_L2SQR, k, d_D, d_I);
        if (st != 1) return st;
    }
    // two-pass form (k > 64, or REID_SELECT_TWO_PASS=1 for A/B): query tiles bound the scratch matrix to ~1 GiB
    const int rows_per = (int)((size_t(1) << 28) / (size_t)nb) > 0 ? (int)((size_t(1) << 28) / (size_t)nb) : 1;
    float* dist;
    const int tile = nq < rows_per ? nq : rows_per;
    REID_TRY(ctx_ws(ctx, "sel.dist", (size_t)tile * nb * 4, (void**)&dist));
    for (int i = 0; i < nq; i += tile) {
        const int m = nq - i < tile ? nq - i : tile;
        REID_TRY(reid_distmat_dev(ctx, d_xq + (size_t)i * d, m, d_xb, nb, d, REID_METRIC_L2SQR, dist));
        REID_TRY(launch_topk_rows(ctx, dist, m, nb, nb, k, d_D + (size_t)i * k, d_I + (size_t)i * k));
    }
    return REID_OK;
}

extern "C" int reid_knn(reid_ctx* ctx, const float* xq, int nq, const float* xb, int nb, int d, int k, float* D, int32_t* I) {
    ARG_CHECK(ctx && xq && xb && D && I && nq >= 0 && nb >= 1 && d >= 1 && k >= 1);
    CTX_GUARD(ctx);
    if (nq == 0) return REID_OK;
    HostIO io{ctx};
    REID_TRY(io.upload(xq, nq, xb, nb, d));
    float* d_D;
    int32_t* d_I;
    REID_TRY(ctx_ws(ctx, "io.knnD", (size_t)nq * k * 4, (void**)&d_D));
    REID_TRY(ctx_ws(ctx, "io.knnI", (size_t)nq * k * 4, (void**)&d_I));
    REID_TRY(reid_knn_dev(ctx, io.dx, nq, io.dy, nb, d, k, d_D, d_I));
    HIP_TRY(hipMemcpyAsync(D, d_D, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(I, d_I, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}

static int reid_diou_cost_impl(reid_ctx* ctx, const double* tracks, int t, const double* dets, int m, double* out,
                                   int as_cost) {
    ARG_CHECK(ctx && tracks && dets && out && t >= 0 && m >= 0);
    CTX_GUARD(ctx);
    if (t == 0 || m == 0) return REID_OK;
    double *dt, *dd, *dout;
    REID_TRY(ctx_ws(ctx, "diou.t", (size_t)t * 32, (void**)&dt));
    REID_TRY(ctx_ws(ctx, "diou.d", (size_t)m * 32, (void**)&dd));
    REID_TRY(ctx_ws(ctx, "diou.o", (size_t)t * m * 8, (void**)&dout));
    HIP_TRY(hipMemcpyAsync(dt, tracks, (size_t)t * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dd, dets, (size_t)m * 32, hipMemcpyHostToDevice, ctx->stream));
    REID_TRY(launch_diou_cost(ctx, dt, t, dd, m, dout, as_cost));
    HIP_TRY(hipMemcpyAsync(out, dout, (size_t)t * m * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}
extern "C" int reid_diou(reid_ctx* ctx, const double* box4, const double* cand, int m, double* out) {
    return reid_diou_cost_impl(ctx, box4, 1, cand, m, out, 0);
}
extern "C" int reid_diou_cost(reid_ctx* ctx, const double* tracks, int t, const double* dets, int m, double* out) {
    return reid_diou_cost_impl(ctx, tracks, t, dets, m, out, 1);
}

// features already on the device (the evaluation harness keeps them there between links); labels and results are host arrays
extern "C" int reid_rank_eval_dev(reid_ctx* ctx, const float* d_qf, const int64_t* ql, const int64_t* qc, int nq,
                                  const float* d_gf, const int64_t* gl, const int64_t* gc, int ng, int d, int32_t* cmc_sum,
                                  double* ap, int32_t* valid) {
    ARG_CHECK(ctx && d_qf && ql && qc && d_gf && gl && gc && cmc_sum && ap && valid && nq >= 1 && ng >= 1 && d >= 1);
    CTX_GUARD(ctx);
    long long *dql, *dqc, *dgl, *dgc;
    int32_t *dhist, *dvalid;
    double* dap;
    float* score;
    REID_TRY(ctx_ws(ctx, "ev.ql", (size_t)nq * 8, (void**)&dql));
    REID_TRY(ctx_ws(ctx, "ev.qc", (size_t)nq * 8, (void**)&dqc));
    REID_TRY(ctx_ws(ctx, "ev.gl", (size_t)ng * 8, (void**)&dgl));
    REID_TRY(ctx_ws(ctx, "ev.gc", (size_t)ng * 8, (void**)&dgc));
    REID_TRY(ctx_ws(ctx, "ev.hist", (size_t)ng * 4, (void**)&dhist));
    REID_TRY(ctx_ws(ctx, "ev.valid", (size_t)nq * 4, (void**)&dvalid));
    REID_TRY(ctx_ws(ctx, "ev.ap", (size_t)nq * 8, (void**)&dap));
    REID_TRY(ctx_ws(ctx, "sel.dist", (size_t)nq * ng * 4, (void**)&score));
    HIP_TRY(hipMemcpyAsync(dql, ql, (size_t)nq * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dqc, qc, (size_t)nq * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dgl, gl, (size_t)ng * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dgc, gc, (size_t)ng * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(dhist, 0, (size_t)ng * 4, ctx->stream));
    REID_TRY(reid_distmat_dev(ctx, d_qf, nq, d_gf, ng, d, REID_METRIC_DOT, score));
    REID_TRY(launch_rank_eval(ctx, score, nq, ng, ng, dql, dqc, dgl, dgc, dhist, dap, dvalid));
    HIP_TRY(hipMemcpyAsync(cmc_sum, dhist, (size_t)ng * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ap, dap, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(valid, dvalid, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int q = 0; q < nq; ++q)
        if (valid[q] < 0) {
            reid_set_error("reid_rank_eval: query %d has more than 2048 good gallery items", q);
            return REID_ERR_ARG;
        }
    // histogram of first-good ranks -> summed CMC step functions (cmc[rows_good[0]:] = 1, evaluate.py:95)
    for (int j = 1; j < ng; ++j) cmc_sum[j] += cmc_sum[j - 1];
    return REID_OK;
}

extern "C" int reid_rank_eval(reid_ctx* ctx, const float* qf, const int64_t* ql, const int64_t* qc, int nq, const float* gf,
                              const int64_t* gl, const int64_t* gc, int ng, int d, int32_t* cmc_sum, double* ap,
                              int32_t* valid) {
    ARG_CHECK(ctx && qf && gf && nq >= 1 && ng >= 1 && d >= 1);
    CTX_GUARD(ctx);
    HostIO io{ctx};
    REID_TRY(io.upload(qf, nq, gf, ng, d));
    return reid_rank_eval_dev(ctx, io.dx, ql, qc, nq, io.dy, gl, gc, ng, d, cmc_sum, ap, valid);
}

// ------------------------------------------------------------------------------------------------ single operators
extern "C" int reid_conv2d_nhwc(reid_ctx* ctx, const float* x, int n, int h, int w, int cin, const float* wgt, int cout, int r,
                                int s, int stride, int pad, const float* scale, const float* shift, const float* residual,
                                int relu, float* out) {
    ARG_CHECK(ctx && x && wgt && out && n >= 1 && cin % 32 == 0 && (scale == nullptr) == (shift == nullptr));
    CTX_GUARD(ctx);
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (w + 2 * pad - s) / stride + 1;
    const size_t nin = (size_t)n * h * w * cin, nw = (size_t)cout * r * s * cin, nout = (size_t)n * ho * wo * cout;
    float *dx, *dw, *dout, *dsc = nullptr, *dsh = nullptr, *dres = nullptr;
    REID_TRY(ctx_ws(ctx, "op.x", nin * 4, (void**)&dx));
    REID_TRY(ctx_ws(ctx, "op.w", nw * 4, (void**)&dw));
    REID_TRY(ctx_ws(ctx, "op.out", nout * 4, (void**)&dout));
    HIP_TRY(hipMemcpyAsync(dx, x, nin * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dw, wgt, nw * 4, hipMemcpyHostToDevice, ctx->stream));
    if (scale) {
        REID_TRY(ctx_ws(ctx, "op.scale", (size_t)cout * 4, (void**)&dsc));
        REID_TRY(ctx_ws(ctx, "op.shift", (size_t)cout * 4, (void**)&dsh));
        HIP_TRY(hipMemcpyAsync(dsc, scale, (size_t)cout * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(dsh, shift, (size_t)cout * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    if (residual) {
        REID_TRY(ctx_ws(ctx, "op.res", nout * 4, (void**)&dres));
        HIP_TRY(hipMemcpyAsync(dres, residual, nout * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    REID_TRY(conv_gemm(ctx, A_IM2COL, dx, n, h, w, cin, dw, cout, r, s, stride, pad, r * s * cin, nullptr, nullptr, 0, dsc, dsh,
                       dres, relu, nullptr, dout));
    HIP_TRY(hipMemcpyAsync(out, dout, nout * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}

extern "C" int reid_gemm_nt(reid_ctx* ctx, const float* a, int m, const float* b, int n, int k, const float* bias, float* c) {
    ARG_CHECK(ctx && a && b && c && m >= 1 && n >= 1 && k >= 1);
    CTX_GUARD(ctx);
    float *da, *db, *dc, *dbias = nullptr;
    REID_TRY(ctx_ws(ctx, "op.x", (size_t)m * k * 4, (void**)&da));
    REID_TRY(ctx_ws(ctx, "op.w", (size_t)n * k * 4, (void**)&db));
    REID_TRY(ctx_ws(ctx, "op.out", (size_t)m * n * 4, (void**)&dc));
    HIP_TRY(hipMemcpyAsync(da, a, (size_t)m * k * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db, b, (size_t)n * k * 4, hipMemcpyHostToDevice, ctx->stream));
    if (bias) {
        REID_TRY(ctx_ws(ctx, "op.shift", (size_t)n * 4, (void**)&dbias));
        HIP_TRY(hipMemcpyAsync(dbias, bias, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    const float *ap, *bp;
    int lda, ldb;
    REID_TRY(pad_rows(ctx, "dist.xpad", da, m, k, &ap, &lda));
    REID_TRY(pad_rows(ctx, "dist.ypad", db, n, k, &bp, &ldb));
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = ap; p.lda = lda;
    p.B = bp; p.ldb = ldb;
    p.M = m; p.N = n; p.K = lda;
    p.C = dc; p.ldc = n;
    p.col_shift = dbias;
    REID_TRY(launch_gemm_f32(ctx, A_DENSE, E_BIAS, p, REID_K_CONV_GEMM, 2.0 * m * n * k,
                             4.0 * ((double)m * k + (double)n * k + (double)m * n)));
    HIP_TRY(hipMemcpyAsync(c, dc, (size_t)m * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}


