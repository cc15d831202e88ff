// fy_itemsim.hip -- item-item similarity build (hot path #2): the co-rating row kernel with a top-K epilogue.
//
// Replaces Mahout 0.8's RowSimilarityJob as called at M/baselinerecommender/BaselineRecommenderJob.java:241-253
// (--similarityClassname, --maxSimilaritiesPerRow, --excludeSelfSimilarity, --threshold).  Parity is UNPINNED: the
// arithmetic is not in the reference tree and no reference test covers it; the algorithm
// restated here is Mahout's published one (see oracle/itemsim_oracle.c for the statement both sides implement):
//   cosine        sim(i, j) = sum_u (r_ui / |r_.i|) (r_uj / |r_.j|)      (rows L2-normalised, then dot products)
//   co-occurrence sim(i, j) = #users who rated both
// over co-rated pairs only, j != i when excludeSelfSimilarity, sim >= threshold (no threshold: sim > 0), the
// maxSimilaritiesPerRow best per item (ties, unspecified in Mahout, by ascending item id).
// No dense I x I matrix is ever written: a workgroup owns item row i, accumulates it chunk by chunk in LDS (fp64) and
// keeps a running top-K in LDS; only I x K rows leave the chip.
#include <algorithm>
#include <cmath>
#include <memory>

#include "fy_cooc.hpp"
#include "fy_prep.hpp"
#include "fy_rm2.hpp"

namespace fy {

static inline int grid_for(int64_t n, int block = 256, int cap = 256 * 16) {
    int64_t g = ceil_div(n, block);
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

// one wave per item (pair): L2 norm of the item's rating column, fixed summation order
__global__ void k_item_norms(int32_t nP, const int32_t* __restrict__ pair_start, const float* __restrict__ csc_r,
                             double* __restrict__ norm) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int32_t p = blockIdx.x * wpb + (threadIdx.x >> 6); p < nP; p += gridDim.x * wpb) {
        double s = 0.0;
        for (int32_t q = pair_start[p] + lane; q < pair_start[p + 1]; q += 64) s += (double)csc_r[q] * (double)csc_r[q];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) norm[p] = sqrt(s);
    }
}

__global__ void k_csc_weights(int64_t nnz, const int32_t* __restrict__ csc_pair, const float* __restrict__ csc_r,
                              const double* __restrict__ norm, int cosine, float* __restrict__ csc_w) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x)
        csc_w[q] = cosine ? (float)((double)csc_r[q] / norm[csc_pair[q]]) : 1.0f;
}

__global__ void k_csr_weights(int64_t nnz, const int32_t* __restrict__ csr_idx, const float* __restrict__ csr_r,
                              const int32_t* __restrict__ rank_pair, const double* __restrict__ norm, int cosine,
                              float* __restrict__ csr_w) {
    for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < nnz; f += (int64_t)gridDim.x * blockDim.x)
        csr_w[f] = cosine ? (float)((double)csr_r[f] / norm[rank_pair[csr_idx[f]]]) : 1.0f;
}

__device__ __forceinline__ uint32_t isim_order_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float isim_order_unkey(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

constexpr int ISIM_CAP = 2048;      // candidate buffer (LDS)
constexpr int ISIM_MAX_K = 1024;

struct ISimEpilogue {
    const int32_t* __restrict__ rank_item_raw;
    int32_t K;
    int32_t exclude_self;
    int32_t has_threshold;
    float threshold;
    int32_t rank, world;     // this launch builds rows rank, rank + world, ...
    int32_t* __restrict__ out_cnt;     // [rows_mine]
    int32_t* __restrict__ out_other;   // [rows_mine * K]
    float* __restrict__ out_sim;       // [rows_mine * K]
};

__device__ __forceinline__ void isim_sort_desc(uint64_t* v, int P2) {
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P2; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const uint64_t x = v[i], y = v[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { v[i] = y; v[l] = x; }
                }
            }
            __syncthreads();
        }
}

// dynamic LDS: [CH doubles accumulators][ISIM_CAP uint64 candidates]
__global__ void k_cooc_itemsim(CoocArgs A, ISimEpilogue E) {
    double* acc = fy_cooc_acc;
    uint64_t* cand = reinterpret_cast<uint64_t*>(fy_cooc_acc + A.CH);
    __shared__ uint32_t sh_cnt, sh_tau;
    const int mine = blockIdx.x;
    const int row = A.row0 + E.rank + mine * E.world;
    const int tid = threadIdx.x;
    if (tid == 0) { sh_cnt = 0; sh_tau = 0; }
    const int self_raw = E.rank_item_raw[row];
    for (int ch = 0; ch < A.nch; ch++) {
        for (int t = tid; t < A.CH; t += blockDim.x) acc[t] = 0.0;
        __syncthreads();
        cooc_accumulate_row(A, row, ch);
        __syncthreads();
        const int c0 = ch * A.CH;
        const int ncol = min(A.CH, A.Ic - c0);
        // stream the finished chunk through the running top-K: keep values >= tau, compact when the buffer fills
        for (int base = 0; base < ncol; base += blockDim.x) {
            const int t = base + tid;
            if (t < ncol) {
                const float s = (float)acc[t];
                const int col = c0 + t;
                bool ok = E.has_threshold ? (s >= E.threshold) : (s > 0.0f);
                if (E.exclude_self && col == row) ok = false;
                if (ok) {
                    const uint32_t key = isim_order_key(s);
                    if (key >= sh_tau) {
                        const uint32_t pos = atomicAdd(&sh_cnt, 1u);
                        cand[pos] = ((uint64_t)key << 32) | (uint32_t)(0x7FFFFFFF - E.rank_item_raw[col]);
                    }
                }
            }
            __syncthreads();
            if (sh_cnt + blockDim.x > (uint32_t)ISIM_CAP) {   // block-uniform: the next step could overflow
                const int n = (int)sh_cnt;
                for (int i = n + tid; i < ISIM_CAP; i += blockDim.x) cand[i] = 0ull;
                __syncthreads();
                isim_sort_desc(cand, ISIM_CAP);
                if (tid == 0) {
                    sh_cnt = (uint32_t)min(n, E.K);
                    if (n >= E.K) sh_tau = (uint32_t)(cand[E.K - 1] >> 32);
                }
                __syncthreads();
            }
        }
        __syncthreads();
    }
    const int n = (int)sh_cnt;
    int P2 = 1;
    while (P2 < n) P2 <<= 1;
    for (int i = n + tid; i < P2; i += blockDim.x) cand[i] = 0ull;
    __syncthreads();
    isim_sort_desc(cand, P2);
    const int keep = min(n, E.K);
    if (tid == 0) E.out_cnt[mine] = keep;
    for (int i = tid; i < keep; i += blockDim.x) {
        const uint64_t c = cand[i];
        E.out_other[(int64_t)mine * E.K + i] = 0x7FFFFFFF - (int32_t)(uint32_t)c;
        E.out_sim[(int64_t)mine * E.K + i] = isim_order_unkey((uint32_t)(c >> 32));
    }
    (void)self_raw;
}

__global__ void k_isim_compact(int32_t rows_mine, int32_t K, int32_t rank, int32_t world, const int32_t* __restrict__ cnt,
                               const int32_t* __restrict__ off, const int32_t* __restrict__ other, const float* __restrict__ sim,
                               const int32_t* __restrict__ rank_item_raw, int32_t* __restrict__ o_item,
                               int32_t* __restrict__ o_other, float* __restrict__ o_sim, int32_t* __restrict__ o_aux) {
    const int wpb = blockDim.x >> 6, lane = threadIdx.x & 63;
    for (int32_t m = blockIdx.x * wpb + (threadIdx.x >> 6); m < rows_mine; m += gridDim.x * wpb) {
        const int32_t item = rank_item_raw[rank + m * world];
        for (int i = lane; i < cnt[m]; i += 64) {
            const int64_t o = (int64_t)off[m] + i;
            o_item[o] = item;
            o_other[o] = other[(int64_t)m * K + i];
            o_sim[o] = sim[(int64_t)m * K + i];
            o_aux[o] = 0;
        }
    }
}

fy_result* itemsim_build(Context* ctx, const fy_itemsim_params* prm, const fy_ratings* R) {
    if (prm->similarity != FY_SIMILARITY_COSINE && prm->similarity != FY_SIMILARITY_COOCCURRENCE)
        FY_FAIL(FY_ERR_INVALID_ARGUMENT, "similarity must be FY_SIMILARITY_COSINE or FY_SIMILARITY_COOCCURRENCE");
    if (prm->max_similarities_per_item <= 0) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "maxSimilaritiesPerRow must be > 0");
    if (prm->max_similarities_per_item > ISIM_MAX_K)
        FY_FAIL(FY_ERR_UNSUPPORTED, "maxSimilaritiesPerRow %d exceeds the kernel limit %d", prm->max_similarities_per_item, ISIM_MAX_K);
    if (prm->world <= 0 || prm->rank < 0 || prm->rank >= prm->world) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "rank %d of world %d", prm->rank, prm->world);
    hipStream_t st = ctx->stream;
    std::unique_ptr<fy_result> Rs(new fy_result);
    Rs->ctx = ctx;
    Rs->kind = 1;
    EventTimer t_prep(ctx), t_cooc(ctx), t_total(ctx);
    const size_t sp0 = t_total.begin();
    const size_t sp1 = t_prep.begin();
    Prepared P;
    build_structure(ctx, R, 1, 0, nullptr, nullptr, nullptr, true, P);
    t_prep.end(sp1);
    Rs->st.nnz = P.nnz;
    Rs->st.n_users = P.nU;
    Rs->st.n_items = P.nI;
    if (P.nnz == 0) {
        t_total.end(sp0);
        sync(ctx);
        return Rs.release();
    }
    const int32_t Ic = P.nP;   // single "cluster": every item is a pair
    const int cosine = prm->similarity == FY_SIMILARITY_COSINE;
    DevBuf<double> norm(ctx, Ic);
    DevBuf<float> csc_w(ctx, P.nnz), csr_w(ctx, P.nnz);
    k_item_norms<<<grid_for((int64_t)Ic * 64, 256), 256, 0, st>>>(Ic, P.pair_start.get(), P.csc_r.get(), norm.get());
    FY_KERNEL_CHECK();
    k_csc_weights<<<grid_for(P.nnz), 256, 0, st>>>(P.nnz, P.csc_pair.get(), P.csc_r.get(), norm.get(), cosine, csc_w.get());
    FY_KERNEL_CHECK();
    k_csr_weights<<<grid_for(P.nnz), 256, 0, st>>>(P.nnz, P.csr_idx.get(), P.csr_r.get(), P.rank_pair.get(), norm.get(), cosine, csr_w.get());
    FY_KERNEL_CHECK();

    const int K = prm->max_similarities_per_item;
    const int max_ch = 16384;   // 128 KiB of fp64 accumulators + 16 KiB candidate buffer <= 160 KiB LDS
    int32_t CH, nch;
    pick_chunks(Ic, max_ch, CH, nch);
    DevBuf<int32_t> chunk_off(ctx, (size_t)P.nU * (nch + 1));
    build_chunk_offsets(ctx, P.rowptr.get(), P.csr_idx.get(), 0, P.nU, CH, nch, chunk_off.get());
    const int32_t rows_mine = (Ic - prm->rank + prm->world - 1) / prm->world;
    DevBuf<int32_t> cnt(ctx, (size_t)rows_mine + 1), off(ctx, (size_t)rows_mine + 1), other(ctx, (size_t)rows_mine * K);
    DevBuf<float> sim(ctx, (size_t)rows_mine * K);
    cnt.zero();
    if (rows_mine > 0) {
        SegTable segs;
        build_segments(ctx, P.csc_slot.get(), csc_w.get(), chunk_off.get(), 0, 0, (int32_t)P.nnz, nch, segs);
        CoocArgs CA{P.rank_pair.get(), P.pair_start.get(), segs.ptr.get(), segs.seg.get(), segs.w.get(), P.csr_idx.get(),
                    csr_w.get(), 0, 0, Ic, CH, nch, 0, Ic, 0, (int32_t)P.nnz, 0};
        ISimEpilogue IE{P.rank_item_raw.get(), K, prm->exclude_self, prm->has_threshold, (float)prm->threshold, prm->rank,
                        prm->world, cnt.get(), other.get(), sim.get()};
        const size_t lds = (size_t)CH * 8 + (size_t)ISIM_CAP * 8;
        FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_itemsim), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int block = lds > 48 * 1024 ? 1024 : 256;
        const size_t sp = t_cooc.begin();
        k_cooc_itemsim<<<rows_mine, block, lds, st>>>(CA, IE);
        FY_KERNEL_CHECK();
        t_cooc.end(sp);
        Rs->st.cooc_launches = 1;
    }
    exclusive_scan_i32(ctx, cnt.get(), off.get(), (size_t)rows_mine + 1);
    const int64_t n = fetch(ctx, off.get() + rows_mine);
    Rs->n = n;
    Rs->d_key0.alloc(ctx, (size_t)n);
    Rs->d_key1.alloc(ctx, (size_t)n);
    Rs->d_value.alloc(ctx, (size_t)n);
    Rs->d_aux.alloc(ctx, (size_t)n);
    if (rows_mine > 0 && n > 0) {
        k_isim_compact<<<grid_for((int64_t)rows_mine * 64, 256), 256, 0, st>>>(rows_mine, K, prm->rank, prm->world, cnt.get(), off.get(),
                                                                               other.get(), sim.get(), P.rank_item_raw.get(),
                                                                               Rs->d_key0.get(), Rs->d_key1.get(), Rs->d_value.get(),
                                                                               Rs->d_aux.get());
        FY_KERNEL_CHECK();
    }
    Rs->d_user_id.alloc(ctx, 0);
    Rs->d_item_id.alloc(ctx, 0);
    t_total.end(sp0);
    sync(ctx);
    Rs->st.recs = n;
    Rs->st.pair_contribs = P.sum_deg2;
    // sum_u n_u (n_u - 1) / 2 = (sum n_u^2 - nnz) / 2: the unit Mahout's co-occurrence mapper enumerates
    Rs->st.unordered_pairs = (P.sum_deg2 - P.nnz) / 2;
    Rs->st.ms_prepare = t_prep.total_ms();
    Rs->st.ms_cooc = t_cooc.total_ms();
    Rs->st.ms_total = t_total.total_ms();
    return Rs.release();
}

}  // namespace fy
