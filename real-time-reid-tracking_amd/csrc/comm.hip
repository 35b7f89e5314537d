// The exchange steps of the multi-GPU path behind the C ABI: one process per GPU, collectives
// straight on librccl (RCCL over xGMI) - no torch.distributed on the data path.
//   * crops are sharded contiguous-by-index, every rank embeds its shard, ONE all-gather of the [n_local, 512] fp32
//     embeddings follows (reid_allgather_dev / reid_allgather_rows_dev), each rank then computes its row block of the matrix;
//   * a fixed gallery (BASELINE config 5) is sharded by rows - the reference's own faiss.IndexShards pattern
//     (reid/faiss_utils.py:121-135: shard, search every shard, merge): reid_knn_gallery_sharded_dev searches this rank's
//     shard, all-gathers the per-shard (distance, index) lists (distances as fp32, indices as int32 - two collectives, no
//     bit-casting of indices through a float payload) and merges them k-way on the device.
// librccl is opened with dlopen on first use: a single-GPU user never loads it.  Without a communicator (world 1) every entry
// point degrades to the local copy, so callers need no special case.
#include "reid_internal.h"
#include <string>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.h) return REID_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        reid_set_error("cannot load librccl.so: %s", dlerror());
        return REID_ERR_STATE;
    }
#define SYM(field, name)                                                   \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                 \
    if (!g_rccl.field) {                                                   \
        reid_set_error("librccl.so lacks %s", name);                       \
        dlclose(h);                                                        \
        return REID_ERR_STATE;                                             \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.h = h;
    return REID_OK;
}

#define RCCL_TRY(expr)                                                                         \
    do {                                                                                       \
        ncclResult_t _r = (expr);                                                              \
        if (_r != ncclSuccess) {                                                               \
            reid_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r)); \
            return REID_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

__device__ __forceinline__ unsigned long long pack_key(float v, int idx) {
    unsigned int u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned int)idx;
}
__device__ __forceinline__ float unpack_val(unsigned long long k) {
    unsigned int u = (unsigned int)(k >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

// k-way merge of the per-shard top-k lists of one query: Dall / Iall are [world][nq][kk] (all-gather order), indices are
// GLOBAL gallery rows (-1 = padding).  Order: ascending distance, ties -> lowest global index (the engine's rule, and what a
// single-process search over the whole gallery returns).  One wave per query; k rounds of "smallest key above the last one".
__global__ __launch_bounds__(64) void knn_merge_kernel(const float* __restrict__ Dall, const int32_t* __restrict__ Iall, int world,
                                                       int nq, int kk, int k, float* __restrict__ D, int32_t* __restrict__ I) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const int cand = world * kk;
    unsigned long long last = 0;
    bool first = true;
    for (int r = 0; r < k; ++r) {
        unsigned long long best = ~0ull;
        for (int c = lane; c < cand; c += 64) {
            const int rk = c / kk, j = c - rk * kk;
            const long long o = ((long long)rk * nq + q) * kk + j;
            const int32_t gi = Iall[o];
            if (gi < 0) continue;
            const unsigned long long key = pack_key(Dall[o], gi);
            if ((first || key > last) && key < best) best = key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o);
            best = other < best ? other : best;
        }
        if (lane == 0) {
            D[(long long)q * k + r] = best == ~0ull ? INFINITY : unpack_val(best);
            I[(long long)q * k + r] = best == ~0ull ? -1 : (int32_t)(best & 0xffffffffu);
        }
        last = best;
        first = false;
    }
}

__global__ void add_index_base_kernel(int32_t* __restrict__ I, long long n, int base) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n && I[i] >= 0) I[i] += base;
}

}  // namespace

int launch_knn_merge(reid_ctx* ctx, const float* Dall, const int32_t* Iall, int world, int nq, int kk, int k, float* D, int32_t* I) {
    hipLaunchKernelGGL(knn_merge_kernel, dim3(nq), dim3(64), 0, ctx->stream, Dall, Iall, world, nq, kk, k, D, I);
    LAUNCH_CHECK();
    return REID_OK;
}

// ------------------------------------------------------------------------------------------------ communicator
extern "C" int reid_comm_unique_id(void* id128) {
    ARG_CHECK(id128);
    REID_TRY(rccl_load());
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == REID_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return REID_OK;
}

extern "C" int reid_comm_init(reid_ctx* ctx, int rank, int world, const void* id128) {
    ARG_CHECK(ctx && world >= 1 && rank >= 0 && rank < world && (world == 1 || id128));
    CTX_GUARD(ctx);
    if (ctx->comm) {
        reid_set_error("reid_comm_init: this context already has a communicator (reid_comm_destroy first)");
        return REID_ERR_STATE;
    }
    reid_comm* c = new reid_comm();
    c->rank = rank;
    c->world = world;
    if (world > 1 || id128) {   // a 1-rank communicator is legal (and exercises the collective code on one GPU)
        if (rccl_load() != REID_OK) { delete c; return REID_ERR_STATE; }
        ncclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        setenv("NCCL_DEBUG", "WARN", 0);   // a failed bring-up must say why (RCCL prints the cause at WARN); never overrides the user
        setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0);   // ... and its version banner and warnings go to stderr: a host's stdout may be parsed
        ncclComm_t nc = nullptr;
        ncclResult_t r = g_rccl.CommInitRank(&nc, world, id, rank);
        if (r != ncclSuccess) {
            // fatal for the caller: there is no second transport.  The usual causes: two ranks on one device ("Duplicate GPU
            // detected", ncclInvalidUsage), a stale / foreign id, ranks that disagree on `world`.
            reid_set_error("ncclCommInitRank(rank %d of %d, device %d) -> %s", rank, world, ctx->device, g_rccl.GetErrorString(r));
            fprintf(stderr, "[libreid_hip] ncclCommInitRank(rank %d of %d, device %d) failed: %s\n", rank, world, ctx->device,
                    g_rccl.GetErrorString(r));
            delete c;
            return REID_ERR_HIP;
        }
        c->comm = nc;
    }
    ctx->comm = c;
    return REID_OK;
}

extern "C" int reid_comm_info(reid_ctx* ctx, int* rank, int* world) {
    ARG_CHECK(ctx);
    if (rank) *rank = ctx->comm ? ctx->comm->rank : 0;
    if (world) *world = ctx->comm ? ctx->comm->world : 1;
    return REID_OK;
}

extern "C" int reid_comm_destroy(reid_ctx* ctx) {
    ARG_CHECK(ctx);
    CTX_GUARD(ctx);
    if (!ctx->comm) return REID_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->comm->loop) ctx->comm->loop->detach(ctx->comm->rank);
    if (ctx->comm->comm) g_rccl.CommDestroy((ncclComm_t)ctx->comm->comm);
    delete ctx->comm;
    ctx->comm = nullptr;
    return REID_OK;
}

void comm_release(reid_ctx* ctx) {   // reid_ctx_destroy
    if (ctx->comm) {
        if (ctx->comm->loop) ctx->comm->loop->detach(ctx->comm->rank);
        if (ctx->comm->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)ctx->comm->comm);
        delete ctx->comm;
        ctx->comm = nullptr;
    }
}

// ------------------------------------------------------------------------------------------------ collectives
// One all-gather of `bytes` bytes per rank on the context's stream: d_recv[r * bytes ..] = rank r's d_send.  No sync.
extern "C" int reid_allgather_dev(reid_ctx* ctx, const void* d_send, void* d_recv, size_t bytes) {
    ARG_CHECK(ctx && d_recv && (d_send || bytes == 0));
    CTX_GUARD(ctx);
    if (bytes == 0) return REID_OK;
    reid_comm* c = ctx->comm;
    if (c && c->loop) return c->loop->allgather(c->rank, d_send, d_recv, bytes, ctx->stream);   // test transport (debug library)
    if (!c || !c->comm) {   // world 1 without a communicator: the gather is a copy
        ARG_CHECK(!c || c->world == 1);
        if (d_recv != d_send) HIP_TRY(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        return REID_OK;
    }
    ncclComm_t nc = (ncclComm_t)c->comm;
    if (bytes % 4 == 0) RCCL_TRY(g_rccl.AllGather(d_send, d_recv, bytes / 4, ncclInt32, nc, ctx->stream));
    else RCCL_TRY(g_rccl.AllGather(d_send, d_recv, bytes, ncclInt8, nc, ctx->stream));
    return REID_OK;
}

// Multi-GPU frames ("per-frame tracking uses the same all-gather"): every rank has submitted its round-robin share
// of a frame's crops to frame slot `slot` (reid_frame_submit); this gathers the ranks' embeddings into the slot as equal blocks
// of `per` = ceil(n / world) rows (ONE ncclAllGather of per x 512 floats per rank; a rank with fewer rows sends padding).
// Afterwards the slot holds world * per rows - row r * per + i is detection r + i * world of the frame - and
// reid_frame_cost / _fetch / _update work on that matrix on every rank.  Asynchronous; a no-op without a communicator.
extern "C" int reid_frame_gather(reid_ctx* ctx, int slot, int per) {
    ARG_CHECK(ctx && (slot == 0 || slot == 1) && per >= 0);
    CTX_GUARD(ctx);
    const int world = ctx->comm ? ctx->comm->world : 1;
    ARG_CHECK(ctx->frame_m[slot] <= per && per <= ctx->frame_m[slot] + 1);   // round-robin shares differ by at most one crop
    if (!ctx->comm || (!ctx->comm->comm && !ctx->comm->loop) || per == 0) return REID_OK;   // (a 1-rank communicator still runs the collective: tests)
    const std::string tag = slot ? "frame1" : "frame0";
    const int mine = ctx->frame_m[slot];
    float *d_loc = ctx->frame_emb[slot], *d_all;
    if (!d_loc || mine == 0) REID_TRY(ctx_ws(ctx, (tag + ".emb").c_str(), (size_t)(per + 1) * 2048, (void**)&d_loc));   // this rank had no crop
    // a rank with one crop fewer than `per` sends a padding row: zeros, not whatever an earlier frame left there (nobody may
    // address it - parallel.frame_rows never does - but a cost matrix over the whole slot must not meet NaNs)
    if (mine < per) HIP_TRY(hipMemsetAsync(d_loc + (size_t)mine * 512, 0, (size_t)(per - mine) * 2048, ctx->stream));
    REID_TRY(ctx_ws(ctx, (tag + ".all").c_str(), (size_t)world * per * 2048, (void**)&d_all));
    const bool two_streams = ctx->match_async && ctx->match_stream;      // reid_frame_match_stream: the slot's readers are on the other stream
    if (two_streams) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->match_ev[slot], 0));
    REID_TRY(reid_allgather_dev(ctx, d_loc, d_all, (size_t)per * 2048));
    if (two_streams) HIP_TRY(hipEventRecord(ctx->fwd_ev[slot], ctx->stream));   // the slot's embeddings are the gathered ones
    ctx->frame_emb[slot] = d_all;
    ctx->frame_m[slot] = world * per;
    ctx->frame_pending[slot] = 1;
    return REID_OK;
}

// Row blocks with possibly different row counts (ragged shards: n % world != 0, fewer crops than ranks): rank r contributes
// n_local rows of row_bytes; d_out receives all rows in rank order, counts_host[world] the per-rank row counts, *n_total their
// sum.  One tiny all-gather for the counts, one for the payload (padded to the largest shard, compacted with D2D copies);
// equal counts skip the padding.  Synchronises the stream once (the counts are needed on the host).
extern "C" int reid_allgather_rows_dev(reid_ctx* ctx, const void* d_local, int n_local, size_t row_bytes, void* d_out,
                                       int32_t* counts_host, int* n_total) {
    ARG_CHECK(ctx && d_out && n_local >= 0 && row_bytes > 0 && (d_local || n_local == 0));
    CTX_GUARD(ctx);
    const int world = ctx->comm ? ctx->comm->world : 1;
    std::vector<int32_t> counts(world, 0);
    if (world == 1) {
        counts[0] = n_local;
    } else {
        int32_t* d_cnt;
        REID_TRY(ctx_ws(ctx, "comm.cnt", (size_t)(world + 1) * 4, (void**)&d_cnt));
        HIP_TRY(hipMemcpyAsync(d_cnt + world, &n_local, 4, hipMemcpyHostToDevice, ctx->stream));
        REID_TRY(reid_allgather_dev(ctx, d_cnt + world, d_cnt, 4));
        HIP_TRY(hipMemcpyAsync(counts.data(), d_cnt, (size_t)world * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    int total = 0, mx = 0;
    bool equal = true;
    for (int r = 0; r < world; ++r) {
        total += counts[r];
        mx = counts[r] > mx ? counts[r] : mx;
        equal = equal && counts[r] == counts[0];
    }
    if (counts_host) memcpy(counts_host, counts.data(), (size_t)world * 4);
    if (n_total) *n_total = total;
    if (total == 0) return REID_OK;
    if (equal) return reid_allgather_dev(ctx, d_local, d_out, (size_t)n_local * row_bytes);
    char *pad, *all;
    const size_t slot = (size_t)mx * row_bytes;
    REID_TRY(ctx_ws(ctx, "comm.pad", slot, (void**)&pad));
    REID_TRY(ctx_ws(ctx, "comm.all", slot * world, (void**)&all));
    if (n_local) HIP_TRY(hipMemcpyAsync(pad, d_local, (size_t)n_local * row_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (n_local < mx) HIP_TRY(hipMemsetAsync(pad + (size_t)n_local * row_bytes, 0, slot - (size_t)n_local * row_bytes, ctx->stream));
    REID_TRY(reid_allgather_dev(ctx, pad, all, slot));
    size_t off = 0;
    for (int r = 0; r < world; ++r) {
        if (counts[r])
            HIP_TRY(hipMemcpyAsync((char*)d_out + off, all + slot * r, (size_t)counts[r] * row_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        off += (size_t)counts[r] * row_bytes;
    }
    return REID_OK;
}

// Small host-side reductions over the ranks (timing: max over ranks; op 0 = sum, 1 = max; count <= 64).  Also the barrier of
// the job: returns after every rank has contributed.  Synchronises the stream.
extern "C" int reid_allreduce_f64(reid_ctx* ctx, double* inout, int count, int op) {
    ARG_CHECK(ctx && inout && count >= 1 && count <= 64 && (op == 0 || op == 1));
    CTX_GUARD(ctx);
    reid_comm* c = ctx->comm;
    if (c && c->loop) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return c->loop->allreduce(c->rank, inout, count, op);
    }
    if (!c || !c->comm) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return REID_OK;
    }
    double* d;
    REID_TRY(ctx_ws(ctx, "comm.red", 64 * 8, (void**)&d));
    HIP_TRY(hipMemcpyAsync(d, inout, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream));
    RCCL_TRY(g_rccl.AllReduce(d, d, count, ncclFloat64, op == 0 ? ncclSum : ncclMax, (ncclComm_t)c->comm, ctx->stream));
    HIP_TRY(hipMemcpyAsync(inout, d, (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return REID_OK;
}

// Squared-L2 k-NN with the GALLERY rows sharded across the ranks (faiss IndexShards pattern, reid/faiss_utils.py:121-135).
// d_xq [nq][d] is the same on every rank, d_xb_local [nb_local][d] this rank's gallery rows, index_base their first global row.
// Every rank ends with the same (d_D float[nq][k] ascending, d_I int32[nq][k] global rows, -1 / +inf when the whole gallery has
// fewer than k rows).  Device-resident: local search, +base, two all-gathers ([nq][k] each), k-way merge kernel.
extern "C" int reid_knn_gallery_sharded_dev(reid_ctx* ctx, const float* d_xq, int nq, const float* d_xb_local, int nb_local,
                                            int index_base, int d, int k, float* d_D, int32_t* d_I) {
    ARG_CHECK(ctx && d_xq && d_D && d_I && nq >= 0 && nb_local >= 0 && index_base >= 0 && d >= 1 && k >= 1);
    CTX_GUARD(ctx);
    if (nq == 0) return REID_OK;
    const int world = ctx->comm ? ctx->comm->world : 1;
    if (world == 1 && index_base > 0) {
        // a shard that does not start at row 0 belongs to a job of several ranks: without a communicator the "merged" result
        // would silently be this shard's alone
        reid_set_error("reid_knn_gallery_sharded_dev: index_base %d > 0 but this context has no communicator of several ranks "
                       "(reid_comm_init): the other shards would be missing from the result", index_base);
        return REID_ERR_STATE;
    }
    float* Dl;
    int32_t* Il;
    const size_t cnt = (size_t)nq * k;
    REID_TRY(ctx_ws(ctx, "comm.knnD", cnt * 4, (void**)&Dl));
    REID_TRY(ctx_ws(ctx, "comm.knnI", cnt * 4, (void**)&Il));
    if (nb_local > 0) {
        ARG_CHECK(d_xb_local);
        REID_TRY(reid_knn_dev(ctx, d_xq, nq, d_xb_local, nb_local, d, k, Dl, Il));   // pads with (+inf, -1) past nb_local
        hipLaunchKernelGGL(add_index_base_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, Il, (long long)cnt,
                           index_base);
        LAUNCH_CHECK();
    } else {   // an empty shard (more ranks than gallery rows) contributes padding only
        HIP_TRY(hipMemsetAsync(Il, 0xff, cnt * 4, ctx->stream));
        HIP_TRY(hipMemsetAsync(Dl, 0x7f, cnt * 4, ctx->stream));   // 0x7f7f7f7f: a large finite float; skipped through index -1
    }
    float* Dall;
    int32_t* Iall;
    REID_TRY(ctx_ws(ctx, "comm.knnDall", cnt * 4 * world, (void**)&Dall));
    REID_TRY(ctx_ws(ctx, "comm.knnIall", cnt * 4 * world, (void**)&Iall));
    REID_TRY(reid_allgather_dev(ctx, Dl, Dall, cnt * 4));
    REID_TRY(reid_allgather_dev(ctx, Il, Iall, cnt * 4));
    return launch_knn_merge(ctx, Dall, Iall, world, nq, k, k, d_D, d_I);
}
