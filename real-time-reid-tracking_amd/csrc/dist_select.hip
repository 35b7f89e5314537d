// Distance GEMM with the SELECTION fused into its epilogue: per row of x the k smallest distances to the rows of y (k = 1: the
// row arg-min) WITHOUT writing the m x n matrix.
//   reid/evaluate.py:58-63        score = gf @ q, argsort, [0]            -> reid_argmin_rows(_dev)
//   reid/faiss_utils.py:97-111    faiss brute-force k-NN (bfKnn)           -> reid_knn(_dev), reid_knn_gallery_sharded_dev
// The two-pass form (gemm_f32_dma_kernel<E_DIST> + topk_rows_kernel) writes and re-reads 4 m n bytes (214 MB for the
// Market-size search, whose operands are 39.5 MB); here a distance leaves the registers only if it can still belong to its
// row's k smallest.
//
// Work split: a block owns 128 rows of x and ONE of S contiguous segments of y's 128-row tiles and sweeps them; the K loop is
// the LDS-DMA loop of gemm_f32_dma.hip (v_mfma_f32_32x32x2_f32, exact fp32 - the distances are bit-identical to the matrix
// kernel's).  Per row the block keeps, in LDS, a threshold thr (the value of the k-th smallest key seen so far in this
// segment; initially the caller's bound, +inf if none) and a counter; an element with v <= thr is appended to the row's
// candidate list in global memory (L2-resident, written by this block only).  Lists hold CAP = 128 keys; a tile is filtered in
// two halves of 64 columns, and a row whose list holds more than 64 keys after a half is compacted to its k smallest (rank by
// counting, one wave per row, in the then idle LDS stage buffers), which also tightens thr - so a list never overflows and
// nothing is lost: the row's true k smallest of the segment are always in its list.  Keys are (order-preserving float bits
// << 32 | column), so "smallest key" = smallest distance, ties -> lowest column, the engine's rule everywhere.
// A second, tiny kernel merges the S sorted lists of a row (select_merge_kernel).
//
// The caller (api.hip) first runs the kernel's BOUND form over a SAMPLE of y (its first 1024 rows): per row the minimum of each
// of k disjoint column groups; the largest of those k minima bounds the row's k-th smallest distance, and a sweep that starts
// from it appends a few per cent of its elements instead of all of them (measured before: a first tile that appends
// everything costs 250 us of list compaction per block).
#include "reid_internal.h"
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = BK * 4;
constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
[[maybe_unused]] constexpr int STAGE = A_BYTES + B_BYTES, AJ = BM / 8 / 4, BJ = BN / 8 / 4;   // device code only
#define LPTR(p) ((__attribute__((address_space(3))) void*)(uintptr_t)(p))
#define RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

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

template <int METRIC>
__device__ __forceinline__ float dist_of(float dot, float rs, float cq) {   // the arithmetic of gemm_f32_dma.hip's dist_epilogue
    if (METRIC == REID_METRIC_L2) return sqrtf(fmaxf(l2sqr_of(dot, rs, cq), 1e-12f));
    if (METRIC == REID_METRIC_L2SQR) return l2sqr_of(dot, rs, cq);
    if (METRIC == REID_METRIC_COS_HALF) return (1.0f - dot / (sqrtf(rs) * cq)) / 2.0f;
    if (METRIC == REID_METRIC_COS) return 1.0f - dot / (sqrtf(rs) * cq);
    return dot;
}

// one wave: the k smallest of list[0..n) (n <= CAP) to dst[0..min(n,k)) in ascending order (dst may be list itself: every key is
// in a register before the first store); returns the new count; the lane that holds the k-th smallest key (if n >= k) gets its
// value in *kth (the others keep theirs).  sc: 128 keys of wave-private LDS.
__device__ __forceinline__ int compact_row(const unsigned long long* list, unsigned long long* dst, int n, int k, unsigned long long* sc,
                                           float* kth, int lane) {
    const unsigned long long k0 = lane < n ? list[lane] : ~0ull;
    const unsigned long long k1 = lane + 64 < n ? list[lane + 64] : ~0ull;
    sc[lane] = k0;
    sc[lane + 64] = k1;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // rank by counting; the keys come as 16-byte broadcast reads, eight per step (slots past n hold ~0: never smaller than a key)
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    int r0 = 0, r1 = 0;
    const int n8 = (n + 7) & ~7;
    for (int j = 0; j < n8; j += 8) {
        const u64x2 a = *(const u64x2*)(sc + j), b = *(const u64x2*)(sc + j + 2), c = *(const u64x2*)(sc + j + 4), d = *(const u64x2*)(sc + j + 6);
        r0 += (a.x < k0) + (a.y < k0) + (b.x < k0) + (b.y < k0) + (c.x < k0) + (c.y < k0) + (d.x < k0) + (d.y < k0);
        r1 += (a.x < k1) + (a.y < k1) + (b.x < k1) + (b.y < k1) + (c.x < k1) + (c.y < k1) + (d.x < k1) + (d.y < k1);
    }
    if (lane < n && r0 < k) dst[r0] = k0;
    if (lane + 64 < n && r1 < k) dst[r1] = k1;
    if (n >= k) {
        if (lane < n && r0 == k - 1) *kth = unpack_val(k0);
        if (lane + 64 < n && r1 == k - 1) *kth = unpack_val(k1);
    }
    __builtin_amdgcn_wave_barrier();
    return n < k ? n : k;
}

__device__ __forceinline__ unsigned int key32(float v) {
    const unsigned int u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unkey32(unsigned int u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// MODE_SELECT: lists + compaction as described at the top.
// MODE_BOUND : no lists - the block takes ONE tile of the sample and folds its distances into p.gmin[row][col % k] (minimum per
//              row and column group).  The k groups are disjoint, so a row's k group minima are k different elements and their
//              maximum bounds the row's k-th smallest distance from above: the select pass starts from that threshold.
// MODE_ARGMIN: k = 1 - no lists either: a candidate goes straight into the row's running minimum (64-bit LDS atomic minimum of
//              the key), which is also the threshold.
enum { MODE_SELECT = 0, MODE_BOUND = 1, MODE_ARGMIN = 2 };

template <int METRIC, int MODE>
__global__ __launch_bounds__(256, 2) void dist_select_kernel(const SelectParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
    __shared__ __attribute__((aligned(16))) float thr[BM];      // value of the row's k-th smallest key so far (exact test)
    __shared__ __attribute__((aligned(16))) float pth[BM];      // REID_METRIC_L2: the same bound before the square root (pre-filter)
    __shared__ __attribute__((aligned(16))) float rsq_s[BM];    // |x_i|^2 of the block's rows
    __shared__ unsigned long long best[MODE == MODE_ARGMIN ? BM : 1];
    __shared__ int cnt[BM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int nk = p.K / BK;                                    // even (K % 64 == 0): every tile starts in LDS stage 0
    const int nmt = (p.M + BM - 1) / BM, nnt = (p.N + BN - 1) / BN;
    const int per = (nnt + p.S - 1) / p.S;

    int t;
    {   // XCD-aware, bijective block remap (blocks b and b+8 share an XCD): an XCD's blocks take consecutive t
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    // t -> (row tile, segment) in 8 x 8 PATCHES: the ~64 blocks an XCD holds at a time (32 CUs x 2) then share 8 tiles of x
    // (2 MB, re-read for every tile of the sweep - from L2 instead of the fabric) and sweep 8 segments of y, each of whose
    // tiles 8 of them read at about the same time.  (Row tile fastest over a whole segment gave an XCD 61 different tiles of x,
    // 15 MB against 4 MB of L2: PMC FETCH_SIZE 0.44 GB per launch for 40 MB of operands.)
    int mtile, seg;
    {
        const int pr = t / (8 * p.S), rem = t - pr * 8 * p.S;
        const int rows_p = nmt - 8 * pr < 8 ? nmt - 8 * pr : 8;
        const int pc = rem / (rows_p * 8), rem2 = rem - pc * rows_p * 8;
        const int si = rem2 / rows_p, mi = rem2 - si * rows_p;
        mtile = 8 * pr + mi;
        seg = 8 * pc + si;
    }
    const int m_blk = mtile * BM;
    const int rows_a = p.M - m_blk < BM ? p.M - m_blk : BM;
    const int nt0 = seg * per, nt1 = nt0 + per < nnt ? nt0 + per : nnt;

    auto set_thr = [&](int r, float v) {   // L2: v <= thr  =>  w <= thr^2 (1 + 1.2e-7) for the correctly rounded sqrtf; 1e-6 covers thr * thr too
        thr[r] = v;
        if (METRIC == REID_METRIC_L2) pth[r] = v * v * (1.0f + 1e-6f);
    };
    if (tid < BM) {
        float t0 = INFINITY;
        if (MODE == MODE_SELECT && p.gmin && tid < rows_a) {   // max over the k group minima of the sample (MODE_BOUND)
            const unsigned int* g = p.gmin + (long long)(m_blk + tid) * p.k;
            unsigned int mx = 0;
            for (int j = 0; j < p.k; ++j) mx = g[j] > mx ? g[j] : mx;
            t0 = mx == 0xffffffffu ? INFINITY : unkey32(mx);
        }
        set_thr(tid, t0);
        rsq_s[tid] = (p.row_sq && tid < rows_a) ? p.row_sq[m_blk + tid] : 0.f;
        cnt[tid] = 0;
        if (MODE == MODE_ARGMIN) best[tid] = ~0ull;
    }
    unsigned long long* lists = p.lists + ((long long)m_blk * p.S + seg) * SEL_CAP;   // row r of the tile: + r * S * CAP
    const long long row_stride = (long long)p.S * SEL_CAP;

    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.A + (long long)m_blk * p.lda), 0, (int)((long long)(rows_a - 1) * p.lda * 4 + (long long)p.K * 4), 0x00020000);
    int a_voff[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = (wave * AJ + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_voff[j] = row < rows_a ? (int)(((long long)row * p.lda + chunk * 4) * 4) : (int)0x7fffff00;
    }
    const int swz = (li >> 1) & 7;
    int a_rd[4], b_rd[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int pos = ((kk * 2 + lh) ^ swz) * 16;
        a_rd[kk] = (wm * 64 + li) * ROWB + pos;
        b_rd[kk] = A_BYTES + (wn * 64 + li) * ROWB + pos;
    }
    // operand descriptor of one tile of y: rows past the last read as zeros (offset beyond the descriptor)
    struct BDesc {
        __amdgpu_buffer_rsrc_t rs;
        int voff[BJ];
    };
    auto describe_b = [&](int nt, BDesc& d) {
        const int n_blk = nt * BN;
        const int rows_b = p.N - n_blk < BN ? p.N - n_blk : BN;
        d.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (long long)n_blk * p.ldb), 0,
                                                 (int)((long long)(rows_b - 1) * p.ldb * 4 + (long long)p.K * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = (wave * BJ + j) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            d.voff[j] = row < rows_b ? (int)(((long long)row * p.ldb + chunk * 4) * 4) : (int)0x7fffff00;
        }
    };
    auto stage = [&](const BDesc& d, int kt, int slot) {
        char* As = lds + slot * STAGE;
        char* Bs = As + A_BYTES;
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, LPTR(As + (wave * AJ + j) * 1024), 16, a_voff[j], kt * BK * 4, 0, 0);
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(d.rs, LPTR(Bs + (wave * BJ + j) * 1024), 16, d.voff[j], kt * BK * 4, 0, 0);
    };
    f32x16 acc[2][2];
    auto mfma_tile = [&](int slot) {
        const char* base = lds + slot * STAGE;
        f32x4 af[2][2], bf[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) af[0][a] = *(const f32x4*)(base + a_rd[0] + a * 32 * ROWB);
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[0][b] = *(const f32x4*)(base + b_rd[0] + b * 32 * ROWB);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) {
#pragma unroll
                for (int a = 0; a < 2; ++a) af[(kk + 1) & 1][a] = *(const f32x4*)(base + a_rd[kk + 1] + a * 32 * ROWB);
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[(kk + 1) & 1][b] = *(const f32x4*)(base + b_rd[kk + 1] + b * 32 * ROWB);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][a][e], bf[kk & 1][b][e], acc[a][b], 0, 0, 0);
        }
    };
    __syncthreads();

    // The K-tiles of the block's successive tiles form ONE stream through the two LDS stages: the last K-tile of a tile has the
    // first K-tile of the next tile in flight behind it (stage 0), so a tile does not start with an exposed DMA round trip; the
    // selection epilogue runs in between, with its scratch in stage 1 (just consumed).
    BDesc cur, nxt;
    describe_b(nt0, cur);
    stage(cur, 0, 0);
    for (int nt = nt0; nt < nt1; ++nt) {
        const int n_blk = nt * BN;
        const bool more = nt + 1 < nt1;
        if (more) describe_b(nt + 1, nxt);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        for (int kt = 0; kt < nk; kt += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of K-tile kt have landed
            RAW_BARRIER();                                     // ... everyone's have, and everyone is done with stage 1
            stage(cur, kt + 1, 1);
            mfma_tile(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RAW_BARRIER();
            if (kt + 2 < nk) stage(cur, kt + 2, 0);
            else if (more) stage(nxt, 0, 0);
            mfma_tile(1);
        }
        RAW_BARRIER();   // every wave has left the K loop: stage 1 is free (scratch); stage 0 is being filled for the next tile
        char* scratch = lds + STAGE;

        if constexpr (MODE == MODE_BOUND) {   // group minima of this tile in LDS, then one global atomic per (row, group)
            unsigned int* gm = (unsigned int*)scratch;
            for (int i = tid; i < BM * p.k; i += 256) gm[i] = 0xffffffffu;
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = n_blk + wn * 64 + b * 32 + li;
                const bool colok = col < p.N;
                float cq = (colok && p.col_sq) ? p.col_sq[col] : 0.f;
                if (METRIC == REID_METRIC_COS_HALF || METRIC == REID_METRIC_COS) cq = sqrtf(cq);
                const int grp = col % p.k;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int rbase = wm * 64 + a * 32 + 4 * lh;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = rbase + (e & 3) + 8 * (e >> 2);
                        const float v = dist_of<METRIC>(acc[a][b][e], rsq_s[row], cq);
                        if (colok && v == v) atomicMin(&gm[row * p.k + grp], key32(v));
                    }
                }
            }
            __syncthreads();
            for (int i = tid; i < rows_a * p.k; i += 256)
                if (gm[i] != 0xffffffffu) atomicMin(p.gmin + (long long)m_blk * p.k + i, gm[i]);
            __syncthreads();
        } else if (p.exp_skip != 1) {
            // ---- filter: two halves of 64 columns (b = MFMA column block of each of the two wave columns)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = n_blk + wn * 64 + b * 32 + li;
                const bool colok = col < p.N;
                float cq = (colok && p.col_sq) ? p.col_sq[col] : 0.f;
                if (METRIC == REID_METRIC_COS_HALF || METRIC == REID_METRIC_COS) cq = sqrtf(cq);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int rbase = wm * 64 + a * 32 + 4 * lh;
                    float th[16], rs[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 t4 = *(const f32x4*)((METRIC == REID_METRIC_L2 ? pth : thr) + rbase + 8 * q);
                        const f32x4 r4 = *(const f32x4*)(rsq_s + rbase + 8 * q);
                        th[4 * q] = t4.x; th[4 * q + 1] = t4.y; th[4 * q + 2] = t4.z; th[4 * q + 3] = t4.w;
                        rs[4 * q] = r4.x; rs[4 * q + 1] = r4.y; rs[4 * q + 2] = r4.z; rs[4 * q + 3] = r4.w;
                    }
                    // L2: pre-filter on the value under the square root (two instructions); the root and the exact test only
                    // for what passes.  The other metrics compare the distance itself.
                    // Two phases per group of 16 elements: all slot requests (LDS atomics with return) are issued before the
                    // first answer is needed, then the keys go out - an `if (pass) { atomic; store }` per element serialised 64
                    // LDS round trips per tile (the body runs whenever ANY lane passes: ~always at a few per cent per lane).
                    float v[16];
                    int slot[16];
                    unsigned pass = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float w = METRIC == REID_METRIC_L2 ? l2sqr_of(acc[a][b][e], rs[e], cq) : dist_of<METRIC>(acc[a][b][e], rs[e], cq);
                        v[e] = w;
                        const int row = rbase + (e & 3) + 8 * (e >> 2);
                        pass |= (colok && row < rows_a && w <= th[e]) ? (1u << e) : 0u;
                    }
                    if (__ballot(pass != 0) == 0) continue;   // nobody in the wave: the common case once thr is tight
                    if (METRIC == REID_METRIC_L2) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            if (pass & (1u << e)) {
                                v[e] = sqrtf(fmaxf(v[e], 1e-12f));
                                if (!(v[e] <= thr[rbase + (e & 3) + 8 * (e >> 2)])) pass &= ~(1u << e);
                            }
                        }
                    }
                    if constexpr (MODE == MODE_ARGMIN) {
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            if (pass & (1u << e)) atomicMin(&best[rbase + (e & 3) + 8 * (e >> 2)], pack_key(v[e], col + p.index_base));
                    } else if (p.exp_skip != 2) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) slot[e] = (pass & (1u << e)) ? atomicAdd(&cnt[rbase + (e & 3) + 8 * (e >> 2)], 1) : 0;
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            if (pass & (1u << e)) lists[(rbase + (e & 3) + 8 * (e >> 2)) * row_stride + slot[e]] = pack_key(v[e], col + p.index_base);
                    }
                }
                __syncthreads();   // the half's candidates are in (and visible: workgroup-scope fence)
                if constexpr (MODE == MODE_ARGMIN) {
                    if (tid < BM && best[tid] != ~0ull) set_thr(tid, unpack_val(best[tid]));
                } else {
                    // ---- compaction of the rows whose list passed 64 keys: wave w looks after rows w*32 .. w*32+31
                    unsigned long long* sc = (unsigned long long*)scratch + wave * SEL_CAP;
                    for (int r = wave * 32; r < wave * 32 + 32; ++r) {
                        const int n = cnt[r];
                        if (n > 64) {
                            float kth = INFINITY;
                            const int m2 = compact_row(lists + r * row_stride, lists + r * row_stride, n, p.k, sc, &kth, lane);
                            kth = __shfl(kth, __builtin_ctzll(__ballot(kth != INFINITY) | (1ull << 63)));
                            if (lane == 0) {
                                cnt[r] = m2;
                                if (kth != INFINITY) set_thr(r, kth);
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }
        cur = nxt;
    }
    // ---- results of the segment: the arg-min key, or the length of the row's candidate list (its k smallest are among them; the
    // merge kernel selects - compacting 128 lists at the end of a block, a few waves on an otherwise finished CU, cost 48 us)
    if constexpr (MODE == MODE_ARGMIN) {
        if (tid < rows_a) p.final_keys[(long long)(m_blk + tid) * p.S + seg] = best[tid];
    } else if constexpr (MODE == MODE_SELECT) {
        if (tid < rows_a) p.counts[(long long)(m_blk + tid) * p.S + seg] = cnt[tid];
    }
#endif
}

// One wave per row: merge of the row's S ascending lists final_keys[row][s][0..k) (padded with ~0) -> D[row][k], I[row][k] ascending,
// padded with (+inf, -1).  The S * k keys sit in registers (NPL per lane, coalesced loads); k rounds of wave minimum, the owner
// retires its key.  Keys are unique (they carry the column), so exactly one lane retires per round.
template <int NPL>
__global__ __launch_bounds__(256) void select_merge_kernel(const unsigned long long* __restrict__ final_keys, int M, int S, int k,
                                                           float* __restrict__ D, int32_t* __restrict__ I) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const unsigned long long* L = final_keys + (long long)row * S * k;
    unsigned long long key[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int c = lane + 64 * i;
        key[i] = c < S * k ? L[c] : ~0ull;
    }
    for (int r = 0; r < k; ++r) {
        unsigned long long mine = key[0];
#pragma unroll
        for (int i = 1; i < NPL; ++i) mine = key[i] < mine ? key[i] : mine;
        unsigned long long best = mine;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o);
            best = other < best ? other : best;
        }
        if (lane == 0) {
            if (D) D[(long long)row * k + r] = best == ~0ull ? INFINITY : unpack_val(best);
            if (I) I[(long long)row * k + r] = best == ~0ull ? -1 : (int32_t)(best & 0xffffffffu);
        }
        if (best != ~0ull) {
#pragma unroll
            for (int i = 0; i < NPL; ++i)
                if (key[i] == best) key[i] = ~0ull;
        }
    }
}

// One wave per row, k > 1: selection of the k smallest among the row's S candidate lists (counts[row][s] unsorted keys each,
// <= SEL_CAP; ~700 keys for the Market-size search).  k rounds of wave minimum are latency-bound (a 64-bit cross-lane minimum
// is six LDS-crossbar round trips: 40 us per row); instead:
//   1. the lists go to LDS back to back, then into registers (32 per lane, static indices);
//   2. pivot P = k-th smallest of the 64 per-lane minima (rank by counting over 64 LDS broadcast reads): at least k keys are
//      <= P, and - the lanes' shares being random subsets - only a few more;
//   3. the keys <= P are compacted into LDS (ballot + prefix count) and ranked by counting; a key of rank r < k is output r.
// More than 128 survivors (adversarial data) or more than 2048 keys: k rounds of "smallest key above the last" over memory.
constexpr int MERGE_LDS = 2048;   // keys per wave held in LDS / registers
__global__ __launch_bounds__(256) void select_merge_lists_kernel(const unsigned long long* __restrict__ lists, const int* __restrict__ counts,
                                                                 int M, int S, int k, float* __restrict__ D, int32_t* __restrict__ I) {
    __shared__ unsigned long long sh[4][MERGE_LDS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    typedef unsigned long long u64;
    const u64* L = lists + (long long)row * S * SEL_CAP;
    const int* cn = counts + (long long)row * S;
    const int mycnt = lane < S ? cn[lane] : 0;             // S <= 32: lane s holds the length of list s
    int total = mycnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    auto put = [&](int r, u64 key) {
        if (D) D[(long long)row * k + r] = key == ~0ull ? INFINITY : unpack_val(key);
        if (I) I[(long long)row * k + r] = key == ~0ull ? -1 : (int32_t)(key & 0xffffffffu);
    };
    u64* shw = sh[wave];
    bool done = false;
    if (total <= MERGE_LDS) {
        int at = 0;
        for (int s0 = 0; s0 < S; s0 += 8) {   // eight lists per step, every load of a step issued before the first is used
            u64 v0[8], v1[8];
            int c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c[u] = s0 + u < S ? __shfl(mycnt, s0 + u) : 0;
                const u64* Ls = L + (long long)(s0 + u) * SEL_CAP;
                v0[u] = lane < c[u] ? Ls[lane] : 0;
                v1[u] = lane + 64 < c[u] ? Ls[lane + 64] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (lane < c[u]) shw[at + lane] = v0[u];
                if (lane + 64 < c[u]) shw[at + lane + 64] = v1[u];
                at += c[u];
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        u64 key[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) key[i] = lane + 64 * i < total ? shw[lane + 64 * i] : ~0ull;
        u64 mn = key[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) mn = key[i] < mn ? key[i] : mn;
        __builtin_amdgcn_wave_barrier();
        // pivot: the k-th smallest lane minimum (all keys when fewer than k lanes hold one)
        shw[lane] = mn;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int rk = 0;
        for (int j = 0; j < 64; ++j) rk += shw[j] < mn ? 1 : 0;
        const int holders = total < 64 ? total : 64;
        const int want = (k < holders ? k : holders) - 1;                // rank of the pivot among the lane minima
        const unsigned long long who = __ballot(mn != ~0ull && rk == want);   // unique: keys are unique
        u64 P = ~0ull;
        if (who) {
            const int src = __builtin_ctzll(who);
            P = ((u64)(unsigned)__shfl((int)(mn >> 32), src) << 32) | (unsigned)__shfl((int)(mn & 0xffffffffu), src);
        }
        __builtin_amdgcn_wave_barrier();
        // survivors -> LDS (the minima were consumed above)
        int base = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i * 64 < total) {
                const bool pass = key[i] <= P && key[i] != ~0ull;
                const unsigned long long bal = __ballot(pass);
                const int pos = base + __popcll(bal & ((1ull << lane) - 1));
                if (pass && pos < 128) shw[pos] = key[i];
                base += __popcll(bal);
            }
        }
        if (base <= 128) {
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const u64 k0 = lane < base ? shw[lane] : ~0ull, k1 = lane + 64 < base ? shw[lane + 64] : ~0ull;
            int r0 = 0, r1 = 0;
            for (int j = 0; j < base; ++j) {
                const u64 kj = shw[j];
                r0 += kj < k0 ? 1 : 0;
                r1 += kj < k1 ? 1 : 0;
            }
            if (lane < base && r0 < k) put(r0, k0);
            if (lane + 64 < base && r1 < k) put(r1, k1);
            for (int r = base + lane; r < k; r += 64) put(r, ~0ull);    // fewer than k candidates in all of y
            done = true;
        }
    }
    if (done) return;
    u64 last = 0;
    bool first = true;
    for (int r = 0; r < k; ++r) {
        u64 best = ~0ull;
        for (int s = 0; s < S; ++s) {
            const int c = __shfl(mycnt, s);
            for (int j = lane; j < c; j += 64) {
                const u64 key = L[(long long)s * SEL_CAP + j];
                if ((first || key > last) && key < best) best = key;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 other = __shfl_xor(best, o);
            best = other < best ? other : best;
        }
        if (lane == 0) put(r, best);
        last = best;
        first = false;
        if (best == ~0ull) last = ~0ull - 1;   // nothing left: every later round finds nothing either
    }
}

template <int METRIC>
void launch_metric(reid_ctx* ctx, const SelectParams& p, int blocks, int mode) {
    if (mode == MODE_BOUND) hipLaunchKernelGGL((dist_select_kernel<METRIC, MODE_BOUND>), dim3(blocks), dim3(256), 0, ctx->stream, p);
    else if (mode == MODE_ARGMIN) hipLaunchKernelGGL((dist_select_kernel<METRIC, MODE_ARGMIN>), dim3(blocks), dim3(256), 0, ctx->stream, p);
    else hipLaunchKernelGGL((dist_select_kernel<METRIC, MODE_SELECT>), dim3(blocks), dim3(256), 0, ctx->stream, p);
}
void launch_any(reid_ctx* ctx, const SelectParams& p, int blocks, int mode) {
    switch (p.metric) {
        case REID_METRIC_L2: launch_metric<REID_METRIC_L2>(ctx, p, blocks, mode); break;
        case REID_METRIC_L2SQR: launch_metric<REID_METRIC_L2SQR>(ctx, p, blocks, mode); break;
        case REID_METRIC_COS_HALF: launch_metric<REID_METRIC_COS_HALF>(ctx, p, blocks, mode); break;
        case REID_METRIC_COS: launch_metric<REID_METRIC_COS>(ctx, p, blocks, mode); break;
        default: launch_metric<REID_METRIC_DOT>(ctx, p, blocks, mode); break;
    }
}

}  // namespace

int select_segments(int m, int n) {
    // blocks = row tiles x S must fit ONE round of two blocks per CU (256 CUs): persistent blocks of a second, partial round would
    // run while most of the chip idles (125 row tiles x 5 segments = 625 blocks took as long as 1024 would).  Never more
    // segments than column tiles, at most 32 (S * k <= 2048 keys: the merge keeps them in registers), no empty segment.
    const int nmt = (m + BM - 1) / BM, nnt = (n + BN - 1) / BN;
    int S = 512 / nmt;
    if (S > nnt) S = nnt;
    if (S > 32) S = 32;
    if (S < 1) S = 1;
    const int per = (nnt + S - 1) / S;
    return (nnt + per - 1) / per;
}

bool dist_select_supported(const SelectParams& p) {
    return p.k >= 1 && p.k <= SEL_KMAX && p.K % (2 * BK) == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0 &&
           (long long)BM * p.lda * 4 < 0x7fff0000ll && (long long)BN * p.ldb * 4 < 0x7fff0000ll && ((uintptr_t)p.A % 16) == 0 &&
           ((uintptr_t)p.B % 16) == 0;
}

// Sample pass: group minima of the first p.N (= sample) rows of y into p.gmin [M][k] (set to 0xff.. by the caller); one tile per block.
int launch_dist_bound(reid_ctx* ctx, const SelectParams& p) {
    const int nmt = (p.M + BM - 1) / BM, nnt = (p.N + BN - 1) / BN;
    SelectParams q = p;
    q.S = nnt;
    prof_begin(ctx, REID_K_DIST_GEMM, 2.0 * p.M * p.N * p.K, 4.0 * ((double)p.M * p.K + (double)p.N * p.K));
    launch_any(ctx, q, nmt * nnt, MODE_BOUND);
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}

// p.lists: [M][S][SEL_CAP] u64 scratch (unused for k = 1), p.final_keys: [M][S][k] u64 scratch, p.gmin: null or the sample's group minima.
int launch_dist_select(reid_ctx* ctx, const SelectParams& p, float* d_D, int32_t* d_I) {
    const int nmt = (p.M + BM - 1) / BM;
    prof_begin(ctx, REID_K_DIST_GEMM, 2.0 * p.M * p.N * p.K, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + 2.0 * p.M * p.k));
    launch_any(ctx, p, nmt * p.S, p.k == 1 ? MODE_ARGMIN : MODE_SELECT);
    prof_end(ctx);
    LAUNCH_CHECK();
    prof_begin(ctx, REID_K_SELECT, 0, (double)p.M * p.S * p.k * 8.0);
    const int slots = p.S * p.k, blocks = (p.M + 3) / 4;
    if (p.k > 1) {
        hipLaunchKernelGGL(select_merge_lists_kernel, dim3(blocks), dim3(256), 0, ctx->stream, p.lists, p.counts, p.M, p.S, p.k, d_D, d_I);
        prof_end(ctx);
        LAUNCH_CHECK();
        return REID_OK;
    }
#define MERGE(NPL) hipLaunchKernelGGL((select_merge_kernel<NPL>), dim3(blocks), dim3(256), 0, ctx->stream, p.final_keys, p.M, p.S, p.k, d_D, d_I)
    if (slots <= 64) MERGE(1);
    else if (slots <= 128) MERGE(2);
    else if (slots <= 256) MERGE(4);
    else if (slots <= 512) MERGE(8);
    else if (slots <= 1024) MERGE(16);
    else MERGE(32);
#undef MERGE
    prof_end(ctx);
    LAUNCH_CHECK();
    return REID_OK;
}
