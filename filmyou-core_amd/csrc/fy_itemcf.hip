// fy_itemcf.hip -- item-based CF recommendation on top of the similarity matrix ("next" row 1).
//
// Replaces the partialMultiply + aggregateAndRecommend jobs of M/baselinerecommender/BaselineRecommenderJob.java:285-328,
// 340-393.  Followed in the reference tree: M/baselinerecommender/BaselineAggregateAndRecommendReducer.java:97-161
// (numerators / denominators / "at least 2 datapoints"), :80-96 (boolean data), :195-235 ((float) cast, NaN skipped, top
// numRecommendations).  The Mahout 0.8 mappers in front of it are third-party and restated from the published
// algorithm (unverified, see oracle/itemcf_oracle.c): strongest maxPrefsPerUser preferences kept (ties at the cut
// kept), the column of item j = its similarity row plus (j, NaN).  PARITY UNPINNED: no reference test covers the package.
//
// Layout: users in batches; per batch dense fp64 numerator / denominator and int32 count rows in HBM (one row per
// user, columns in popularity order).  One wave per user walks the user's kept preferences in order and adds the
// similarity row of each (<= maxSimilaritiesPerRow entries, one per lane) with L2 atomics, so the per-cell order of
// additions is the order of the preferences; a second kernel turns the cells into float predictions (NaN = skip) and the
// top-N kernels of the RM2 job pick the lists.
#include <algorithm>
#include <cmath>
#include <memory>

#include "fy_prep.hpp"
#include "fy_rm2.hpp"

namespace fy {

static inline int grid_for(int64_t n, int block = 256, int cap = 256 * 16) {
    int64_t g = ceil_div(n, block);
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

// raw item id -> compact column (popularity rank); -1 when the item has no rating
__device__ __forceinline__ int32_t icf_column(const int32_t* __restrict__ iid, int32_t nI, const int32_t* __restrict__ pair_rank, int32_t raw) {
    int32_t lo = 0, hi = nI;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (iid[mid] < raw) lo = mid + 1; else hi = mid;
    }
    return (lo < nI && iid[lo] == raw) ? pair_rank[lo] : -1;
}

// similarity rows (grouped by item) -> per-column row start / length, entries re-expressed as columns
__global__ void k_icf_index_sims(int64_t n, const int32_t* __restrict__ s_item, const int32_t* __restrict__ s_other,
                                 const int32_t* __restrict__ iid, int32_t nI, const int32_t* __restrict__ pair_rank,
                                 int32_t* __restrict__ col_other, int32_t* __restrict__ row_start, int32_t* __restrict__ row_cnt) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        col_other[t] = icf_column(iid, nI, pair_rank, s_other[t]);
        const int32_t it = s_item[t];
        const int32_t row = icf_column(iid, nI, pair_rank, it);
        if (row < 0) continue;
        if (t == 0 || s_item[t - 1] != it) row_start[row] = (int32_t)t;
        atomicAdd(&row_cnt[row], 1);
    }
}

__device__ __forceinline__ uint32_t icf_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// one wave per user: threshold = the maxPrefs-th largest preference value when the user has more (else -inf):
// radix select on the order-preserving integer image, four 8-bit passes with a per-wave LDS histogram
__global__ void k_icf_threshold(int32_t lo, int32_t hi, const int32_t* __restrict__ rowptr, const float* __restrict__ csr_r,
                                int32_t max_prefs, float* __restrict__ thr) {
    __shared__ uint32_t hist[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    for (int32_t s = lo + blockIdx.x * wpb + w; s < hi; s += gridDim.x * wpb) {
        const int32_t a = rowptr[s], b = rowptr[s + 1];
        if (b - a <= max_prefs) {
            if (lane == 0) thr[s - lo] = -INFINITY;
            continue;
        }
        uint32_t prefix = 0, mask = 0, need = (uint32_t)max_prefs;
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 24 - 8 * pass;
            for (int x = lane; x < 256; x += 64) hist[w][x] = 0;
            __builtin_amdgcn_wave_barrier();
            for (int32_t f = a + lane; f < b; f += 64) {
                const uint32_t key = icf_key(csr_r[f]);
                if ((key & mask) == prefix) atomicAdd(&hist[w][(key >> shift) & 255u], 1u);
            }
            __builtin_amdgcn_wave_barrier();
            // lane 0 scans the 256 bins from the top
            uint32_t chosen = 0, cum = 0;
            if (lane == 0) {
                int x = 255;
                for (; x > 0; x--) {
                    if (cum + hist[w][x] >= need) break;
                    cum += hist[w][x];
                }
                chosen = (uint32_t)x;
            }
            chosen = __builtin_amdgcn_readfirstlane(chosen);
            cum = __builtin_amdgcn_readfirstlane(cum);
            prefix |= chosen << shift;
            mask |= 255u << shift;
            need -= cum;
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) {
            const uint32_t bkey = (prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix;
            thr[s - lo] = __uint_as_float(bkey);
        }
    }
}

struct IcfArgs {
    const int32_t* __restrict__ rowptr;
    const int32_t* __restrict__ csr_idx;
    const float* __restrict__ csr_r;
    const float* __restrict__ thr;         // by slot - slot_lo
    const int32_t* __restrict__ row_start;
    const int32_t* __restrict__ row_cnt;
    const int32_t* __restrict__ col_other;
    const float* __restrict__ sim;
    int32_t slot_lo, slot0, n_users, boolean_data;
    int64_t ld;
    double* __restrict__ num;
    double* __restrict__ den;
    int32_t* __restrict__ cnt;
};

__global__ void k_icf_accumulate(IcfArgs A) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int32_t u = blockIdx.x * wpb + (threadIdx.x >> 6); u < A.n_users; u += gridDim.x * wpb) {
        const int32_t slot = A.slot0 + u;
        const float thr = A.thr[slot - A.slot_lo];
        double* __restrict__ num = A.num + (int64_t)u * A.ld;
        double* __restrict__ den = A.den + (int64_t)u * A.ld;
        int32_t* __restrict__ cnt = A.cnt + (int64_t)u * A.ld;
        for (int32_t f = A.rowptr[slot]; f < A.rowptr[slot + 1]; f++) {   // preferences in a fixed order
            const float p = A.csr_r[f];
            if (p < thr) continue;
            const int32_t j = A.csr_idx[f];
            const int32_t n = A.row_cnt[j];
            if (n == 0) continue;                                          // no similarity row: contributes nothing
            const int32_t t0 = A.row_start[j];
            const double pd = A.boolean_data ? 1.0 : (double)p;
            for (int32_t t = t0 + lane; t < t0 + n; t += 64) {
                const int32_t i = A.col_other[t];
                if (i < 0) continue;
                const double s = (double)A.sim[t];
                atomicAdd(&num[i], pd * s);
                atomicAdd(&den[i], fabs(s));
                atomicAdd(&cnt[i], 1);
            }
            if (lane == 0) {   // the (j, NaN) entry of the wrapped similarity column: the item itself drops out
                atomicAdd(&num[j], __builtin_nan(""));
                atomicAdd(&den[j], __builtin_nan(""));
                atomicAdd(&cnt[j], 1);
            }
        }
    }
}

// cells -> float predictions (NaN = not recommendable); valid predictions counted per user
__global__ void k_icf_finalize(int32_t n_users, int32_t n_cols, int64_t ld, const double* __restrict__ num,
                               const double* __restrict__ den, const int32_t* __restrict__ cnt, int32_t boolean_data,
                               int32_t top_n, float* __restrict__ S, int32_t* __restrict__ n_out) {
    const int u = blockIdx.x;
    __shared__ int sh_valid;
    if (threadIdx.x == 0) sh_valid = 0;
    __syncthreads();
    int valid = 0;
    const float qnan = __builtin_nanf("");
    for (int i = threadIdx.x; i < (int)ld; i += blockDim.x) {
        float v = qnan;
        if (i < n_cols) {
            const int64_t o = (int64_t)u * ld + i;
            const int32_t c = cnt[o];
            // The reducer walks numerators.nonZeroes() and recommendationVector.nonZeroes()
            // (BaselineAggregateAndRecommendReducer.java:148, 195): a cell whose numerator is exactly 0 (similarities of both
            // signs that cancel, or 0-valued preferences) never reaches the top-N queue.
            const double nm = num[o];
            if (nm != 0.0) {
                if (boolean_data) { if (c > 0) v = (float)nm; }
                else if (c > 1) v = (float)(nm / den[o]);
            }
        }
        S[(int64_t)u * ld + i] = v;
        valid += (v == v);
    }
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_down(valid, o, 64);
    if ((threadIdx.x & 63) == 0 && valid) atomicAdd(&sh_valid, valid);
    __syncthreads();
    if (threadIdx.x == 0) n_out[u] = min(top_n, sh_valid);
}

__global__ void k_icf_offsets(int32_t n_users, int32_t top_n, int32_t* __restrict__ off) {
    for (int32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n_users; u += gridDim.x * blockDim.x) off[u] = u * top_n;
}

__global__ void k_icf_compact(int32_t n_users, int32_t top_n, const int32_t* __restrict__ cnt, const int32_t* __restrict__ off,
                              const int32_t* __restrict__ p_user, const int32_t* __restrict__ p_item, const float* __restrict__ p_score,
                              int32_t* __restrict__ o_user, int32_t* __restrict__ o_item, float* __restrict__ o_score,
                              int32_t* __restrict__ o_aux) {
    const int wpb = blockDim.x >> 6, lane = threadIdx.x & 63;
    for (int32_t u = blockIdx.x * wpb + (threadIdx.x >> 6); u < n_users; u += gridDim.x * wpb)
        for (int i = lane; i < cnt[u]; i += 64) {
            const int64_t src = (int64_t)u * top_n + i, dst = (int64_t)off[u] + i;
            o_user[dst] = p_user[src];
            o_item[dst] = p_item[src];
            o_score[dst] = p_score[src];
            o_aux[dst] = 0;
        }
}

fy_result* itemcf_recommend(Context* ctx, const fy_itemcf_params* prm, const fy_ratings* R, fy_result* sims) {
    if (prm->num_recommendations <= 0) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "numRecommendations must be > 0");
    if (prm->num_recommendations > 2048) FY_FAIL(FY_ERR_UNSUPPORTED, "numRecommendations %d exceeds the top-N kernel limit 2048", prm->num_recommendations);
    if (prm->max_prefs_per_user <= 0) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "maxPrefsPerUser must be > 0");
    if (prm->world <= 0 || prm->rank < 0 || prm->rank >= prm->world) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "rank %d of world %d", prm->rank, prm->world);
    hipStream_t st = ctx->stream;
    std::unique_ptr<fy_result> Rs(new fy_result);
    Rs->ctx = ctx;
    Rs->kind = 2;
    EventTimer t_total(ctx), t_prep(ctx);
    const size_t sp0 = t_total.begin();
    const size_t sp1 = t_prep.begin();
    Prepared P;
    build_structure(ctx, R, 1, 0, nullptr, nullptr, nullptr, true, P);
    t_prep.end(sp1);
    Rs->st.nnz = P.nnz;
    Rs->st.n_users = P.nU;
    Rs->st.n_items = P.nI;
    Rs->d_user_id.alloc(ctx, 0);
    Rs->d_item_id.alloc(ctx, 0);
    if (P.nnz == 0) {
        t_total.end(sp0);
        sync(ctx);
        return Rs.release();
    }
    const int32_t nI = P.nP, N = prm->num_recommendations;
    const int64_t n_sim = sims->n;
    // ---- similarity rows by column
    DevBuf<int32_t> col_other(ctx, (size_t)n_sim), row_start(ctx, (size_t)nI), row_cnt(ctx, (size_t)nI);
    row_start.zero();
    row_cnt.zero();
    if (n_sim > 0) {
        k_icf_index_sims<<<grid_for(n_sim), 256, 0, st>>>(n_sim, sims->d_key0.get(), sims->d_key1.get(), P.iid.get(), P.nI,
                                                          P.pair_rank.get(), col_other.get(), row_start.get(), row_cnt.get());
        FY_KERNEL_CHECK();
    }
    // ---- this rank's users: a contiguous range of the (degree-sorted) slot order
    int32_t lo = 0, hi = P.nU;
    if (prm->world > 1) {
        lo = (int32_t)((int64_t)P.nU * prm->rank / prm->world);
        hi = (int32_t)((int64_t)P.nU * (prm->rank + 1) / prm->world);
    }
    const int32_t nmine = hi - lo;
    DevBuf<float> thr(ctx, (size_t)nmine + 1);
    if (nmine > 0) {
        k_icf_threshold<<<grid_for((int64_t)nmine * 64, 256), 256, 0, st>>>(lo, hi, P.rowptr.get(), P.csr_r.get(), prm->max_prefs_per_user, thr.get());
        FY_KERNEL_CHECK();
    }
    // ---- batches of users with dense accumulators
    const int64_t ld = round_up(nI, 256);
    const int64_t per_user = ld * (8 + 8 + 4 + 4);
    int64_t B = std::max<int64_t>(1, ((int64_t)8 << 30) / per_user);
    B = std::min<int64_t>(B, std::max<int32_t>(nmine, 1));
    DevBuf<double> num(ctx, (size_t)(B * ld)), den(ctx, (size_t)(B * ld));
    DevBuf<int32_t> cnt(ctx, (size_t)(B * ld));
    DevBuf<float> S(ctx, (size_t)(B * ld));
    DevBuf<int32_t> n_out(ctx, (size_t)nmine + 1), pad_off(ctx, (size_t)nmine + 1), overflow(ctx, (size_t)B), any_overflow(ctx, 1);
    DevBuf<int32_t> p_user(ctx, (size_t)nmine * N), p_item(ctx, (size_t)nmine * N);
    DevBuf<float> p_score(ctx, (size_t)nmine * N);
    DevBuf<int32_t> p_aux(ctx, (size_t)nmine * N);
    n_out.zero();
    if (nmine > 0) {
        k_icf_offsets<<<grid_for(nmine), 256, 0, st>>>(nmine, N, pad_off.get());
        FY_KERNEL_CHECK();
    }
    for (int32_t s0 = lo; s0 < hi; s0 += (int32_t)B) {
        const int32_t nb = (int32_t)std::min<int64_t>(B, hi - s0);
        FY_HIP(hipMemsetAsync(num.get(), 0, (size_t)nb * ld * 8, st));
        FY_HIP(hipMemsetAsync(den.get(), 0, (size_t)nb * ld * 8, st));
        FY_HIP(hipMemsetAsync(cnt.get(), 0, (size_t)nb * ld * 4, st));
        IcfArgs A{P.rowptr.get(), P.csr_idx.get(), P.csr_r.get(), thr.get(), row_start.get(), row_cnt.get(), col_other.get(),
                  sims->d_value.get(), lo, s0, nb, prm->boolean_data, ld, num.get(), den.get(), cnt.get()};
        k_icf_accumulate<<<grid_for((int64_t)nb * 64, 256, 256 * 32), 256, 0, st>>>(A);
        FY_KERNEL_CHECK();
        k_icf_finalize<<<nb, 256, 0, st>>>(nb, nI, ld, num.get(), den.get(), cnt.get(), prm->boolean_data, N, S.get(), n_out.get() + (s0 - lo));
        FY_KERNEL_CHECK();
        launch_topn_rows(ctx, st, S.get(), ld, nI, nb, n_out.get() + (s0 - lo), pad_off.get() + (s0 - lo), P.rank_item_raw.get(),
                         P.slot2du.get(), P.uid.get(), s0, 0, p_user.get(), p_item.get(), p_score.get(), p_aux.get(),
                         overflow.get(), any_overflow.get());
    }
    // ---- compact the padded lists
    DevBuf<int32_t> off(ctx, (size_t)nmine + 1);
    exclusive_scan_i32(ctx, n_out.get(), off.get(), (size_t)nmine + 1);
    const int64_t n = nmine > 0 ? (int64_t)fetch(ctx, off.get() + nmine) : 0;
    Rs->n = n;
    Rs->d_key0.alloc(ctx, (size_t)n);
    Rs->d_key1.alloc(ctx, (size_t)n);
    Rs->d_value.alloc(ctx, (size_t)n);
    Rs->d_aux.alloc(ctx, (size_t)n);
    if (n > 0) {
        k_icf_compact<<<grid_for((int64_t)nmine * 64, 256), 256, 0, st>>>(nmine, N, n_out.get(), off.get(), p_user.get(), p_item.get(),
                                                                          p_score.get(), Rs->d_key0.get(), Rs->d_key1.get(),
                                                                          Rs->d_value.get(), Rs->d_aux.get());
        FY_KERNEL_CHECK();
    }
    t_total.end(sp0);
    sync(ctx);
    Rs->st.recs = n;
    Rs->st.users_scored = nmine;
    Rs->st.ms_prepare = t_prep.total_ms();
    Rs->st.ms_total = t_total.total_ms();
    return Rs.release();
}

}  // namespace fy
