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
constexpr int ISIM_SAMPLE = 256;       // columns sampled for the first threshold guess of a row
constexpr int ISIM_SAMPLE_RANK = 5;    // ... whose 5th largest is the guess (expected: ~5 * columns / 256 values above it)

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
    // packed row kernel: the accumulators hold sum_v r_vi r_vj (exact in fp64 for fp16-exact ratings); cosine = that times
    // inv_norm[i] inv_norm[j] (rank order).  nullptr: the weights were divided by the norms beforehand.
    const double* __restrict__ inv_norm;
    // per (row, chunk) item: its top K as (order key << 32 | ~raw item id), descending
    int32_t* __restrict__ part_cnt;    // [rows_mine * nch]
    uint64_t* __restrict__ part;       // [rows_mine * nch * K]
    int32_t heavy_rows;                // leading rows of the launch that are split by chunk
    int32_t n_items;                   // heavy_rows * nch + (rows - heavy_rows)
    int32_t cap;                       // candidate buffer entries in LDS (power of two, >= 2 K, <= ISIM_CAP)
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

// Cuts the n candidates in LDS down to (at least) the K best without sorting them: two 256-bin histogram levels over the
// order keys (bits 31..24, then 23..16) locate a 16-bit key prefix T with  #(key >= T) >= K  and  #(key >= T + 1 prefix) < K;
// everything below T goes.  ~8 barriers instead of the 66 of a 2048-element bitonic sort (rocprof: the sorts were half
// of the kernel).  Returns the new count through sh_cnt and the new threshold through sh_tau; all threads call it.
// Ties inside the last prefix all stay, so the result may hold more than K entries -- if it would not fit behind the next
// streaming step the caller falls back to the exact sort.
__device__ __forceinline__ void isim_select(uint64_t* cand, int n, int K, uint32_t* hist, uint32_t* sh_cnt, uint32_t* sh_tau,
                                            uint32_t* sh_aux) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (n <= K) {   // block-uniform: nothing to cut
        return;
    }
    uint32_t prefix = 0, above = 0;
    for (int level = 0; level < 2; level++) {
        const int shift = level == 0 ? 24 : 16;
        for (int b = tid; b < 256; b += nt) hist[b] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            const uint32_t key = (uint32_t)(cand[i] >> 32);
            if (level == 0 || (key >> 24) == (prefix >> 24)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t cum = above;
            int b = 255;
            for (; b > 0; b--) {
                if (cum + hist[b] >= (uint32_t)K) break;
                cum += hist[b];
            }
            sh_aux[0] = prefix | ((uint32_t)b << shift);
            sh_aux[1] = cum;
        }
        __syncthreads();
        prefix = sh_aux[0];
        above = sh_aux[1];
        __syncthreads();
    }
    // keep key >= prefix (in place: all reads happen before the first write)
    uint64_t mine[ISIM_CAP / 256];
    int have = 0;
    for (int i = tid; i < n; i += nt) {
        const uint64_t c = cand[i];
        if ((uint32_t)(c >> 32) >= prefix && have < ISIM_CAP / 256) mine[have++] = c;
    }
    __syncthreads();
    if (tid == 0) *sh_cnt = 0;
    __syncthreads();
    for (int k = 0; k < have; k++) cand[atomicAdd(sh_cnt, 1u)] = mine[k];
    if (tid == 0) *sh_tau = prefix;
    __syncthreads();
}

// buffer (nearly) full: keep the K best (plus ties inside the last key prefix); exact sort when even that does not make room
__device__ __forceinline__ void isim_cut(uint64_t* cand, int K, int cap, uint32_t* hist, uint32_t* sh_cnt, uint32_t* sh_tau, uint32_t* sh_aux) {
    const int tid = threadIdx.x;
    isim_select(cand, (int)*sh_cnt, K, hist, sh_cnt, sh_tau, sh_aux);
    if (*sh_cnt + min((uint32_t)blockDim.x, (uint32_t)cap / 2) > (uint32_t)cap) {   // massive ties inside one key prefix (block-uniform)
        const int n = (int)*sh_cnt;
        __syncthreads();
        for (int i = n + tid; i < cap; i += blockDim.x) cand[i] = 0ull;
        __syncthreads();
        isim_sort_desc(cand, cap);
        if (tid == 0) {
            *sh_cnt = (uint32_t)min(n, K);
            if (n >= K) *sh_tau = (uint32_t)(cand[K - 1] >> 32);   // inclusive: a later tie with a smaller item id still wins
        }
        __syncthreads();
    }
}

// dynamic LDS: [CH doubles accumulators][ISIM_CAP uint64 candidates].  Persistent workgroups pull (row, column chunk) ITEMS
// from a global counter (heavy rows first) and leave the top K of that chunk; k_isim_merge folds a row's chunks together.
// Only the heaviest rows are split by chunk (one workgroup per whole row left the heaviest row -- 10^5 raters, 5e7 slice
// entries -- on a single CU for half of the kernel's run time); a light row is one item and carries its threshold from chunk
// to chunk (splitting every row cost more in top-K work than it gained: 52 ms against 30).  The pass that streams a finished chunk through the top-K re-zeroes the accumulators it reads.
template <bool PK>
__global__ void k_cooc_itemsim(CoocArgs A, ISimEpilogue E, int* __restrict__ next_row) {
    double* acc = fy_cooc_acc;
    uint64_t* cand = reinterpret_cast<uint64_t*>(fy_cooc_acc + A.CH);
    __shared__ uint32_t sh_cnt, sh_tau, sh_aux[2], hist[256];
    __shared__ int sh_row;
    const int tid = threadIdx.x;
    const int stride = A.row_stride ? A.row_stride : 1;
    for (int t = tid; t < A.CH; t += blockDim.x) acc[t] = 0.0;
    int next = 0;
    if (tid == 0) next = atomicAdd(next_row, 1);
    for (;;) {
        if (tid == 0) { sh_row = next; sh_cnt = 0; sh_tau = 0; }
        __syncthreads();
        const int item = sh_row;
        if (item >= E.n_items) break;
        if (tid == 0) next = atomicAdd(next_row, 1);   // the round trip hides behind this item's work
        // the first heavy_rows rows are split into one item per column chunk, every other row is one item
        const bool split = item < E.heavy_rows * A.nch;
        const int mine = split ? item / A.nch : E.heavy_rows + (item - E.heavy_rows * A.nch);
        const int ch_begin = split ? item - mine * A.nch : 0, ch_end = split ? ch_begin + 1 : A.nch;
        const int slot = mine * A.nch + ch_begin;     // where this item's list goes
        const int row = A.row0 + mine * stride;
        const double scale_row = E.inv_norm ? E.inv_norm[row] : 1.0;
        for (int ch = ch_begin; ch < ch_end; ch++) {
            cooc_accumulate_row<PK>(A, row, ch, mine);
            __syncthreads();
            const int c0 = ch * A.CH;
            const int ncol = min(A.CH, A.Ic - c0);
            // Stream the finished chunk through the running top-K: values >= tau are appended to the candidate buffer, every
            // accumulator that has been looked at is re-zeroed.  The whole chunk is taken WITHOUT a barrier (one barrier per
            // 1024 columns was a third of the kernel); a candidate that finds the buffer full leaves its accumulator in place,
            // the buffer is cut back (isim_select raises tau) and the pass runs again over what is left.  A consumed entry
            // reads as similarity 0, which only a threshold <= 0 would accept again: that configuration steps with barriers.
            const bool stepwise = E.has_threshold && !(E.threshold > 0.0f);
            if (!stepwise && sh_tau == 0 && ncol >= 4 * ISIM_SAMPLE) {
                // No threshold yet: thousands of non-zero similarities would flood the candidate buffer and be cut back in
                // several rounds.  Guess one from a sample instead: the ISIM_SAMPLE_RANK-th largest of ISIM_SAMPLE columns
                // spread over the chunk (wave 0, shuffles only), count how many columns reach it (one sweep, no writes) and
                // adopt it if at least K do -- then it is a valid lower bound of the K-th best and the append pass below
                // takes a few hundred candidates in one go.  A bad guess changes nothing: the old path runs.
                if (tid < 64) {
                    uint32_t k4[ISIM_SAMPLE / 64];
#pragma unroll
                    for (int x = 0; x < ISIM_SAMPLE / 64; x++) {
                        const int t = (int)(((int64_t)(tid * (ISIM_SAMPLE / 64) + x) * ncol) / ISIM_SAMPLE);
                        const int col = c0 + t;
                        const double a = acc[t];
                        const float sv = E.inv_norm ? (float)(a * scale_row * E.inv_norm[col]) : (float)a;
                        bool ok = E.has_threshold ? (sv >= E.threshold) : (sv > 0.0f);
                        if (E.exclude_self && col == row) ok = false;
                        k4[x] = ok ? isim_order_key(sv) : 0u;
                    }
                    uint32_t best = 0;
                    for (int r = 0; r < ISIM_SAMPLE_RANK; r++) {
                        uint32_t m = 0;
#pragma unroll
                        for (int x = 0; x < ISIM_SAMPLE / 64; x++) m = max(m, k4[x]);
                        uint32_t wm = m;
                        for (int o = 32; o > 0; o >>= 1) wm = max(wm, (uint32_t)__shfl_xor((int)wm, o, 64));
                        best = wm;
                        // remove one occurrence of the maximum (the first lane that holds it)
                        const unsigned long long holders = __ballot(m == wm && wm != 0);
                        if (holders && tid == __ffsll((long long)holders) - 1) {
                            bool done = false;
#pragma unroll
                            for (int x = 0; x < ISIM_SAMPLE / 64; x++)
                                if (!done && k4[x] == wm) { k4[x] = 0; done = true; }
                        }
                    }
                    if (tid == 0) { sh_aux[0] = best; sh_aux[1] = 0; }
                }
                __syncthreads();
                const uint32_t guess = sh_aux[0];
                if (guess != 0) {   // block-uniform
                    int mine_cnt = 0;
                    for (int t = tid; t < ncol; t += blockDim.x) {
                        const int col = c0 + t;
                        const double a = acc[t];
                        const float sv = E.inv_norm ? (float)(a * scale_row * E.inv_norm[col]) : (float)a;
                        bool ok = E.has_threshold ? (sv >= E.threshold) : (sv > 0.0f);
                        if (E.exclude_self && col == row) ok = false;
                        mine_cnt += ok && isim_order_key(sv) >= guess;
                    }
                    for (int o = 32; o > 0; o >>= 1) mine_cnt += __shfl_down(mine_cnt, o, 64);
                    if ((tid & 63) == 0 && mine_cnt) atomicAdd(&sh_aux[1], (uint32_t)mine_cnt);
                    __syncthreads();
                    if (tid == 0 && sh_aux[1] >= (uint32_t)E.K) sh_tau = guess;
                }
                __syncthreads();
            }
            for (bool again = true; again;) {
                const uint32_t tau = sh_tau;
                for (int base = 0; base < ncol; base += blockDim.x) {
                    const int t = base + tid;
                    bool want = false;
                    uint64_t c = 0;
                    if (t < ncol) {
                        const double a = acc[t];
                        const int col = c0 + t;
                        const float s = E.inv_norm ? (float)(a * scale_row * E.inv_norm[col]) : (float)a;
                        bool ok = E.has_threshold ? (s >= E.threshold) : (s > 0.0f);
                        if (E.exclude_self && col == row) ok = false;
                        const uint32_t key = isim_order_key(s);
                        want = ok && key >= (stepwise ? sh_tau : tau);
                        if (want) c = ((uint64_t)key << 32) | (uint32_t)(0x7FFFFFFF - E.rank_item_raw[col]);
                        else acc[t] = 0.0;
                    }
                    // one LDS atomic per wave
                    const unsigned long long bal = __ballot(want);
                    if (bal) {
                        const int lane = tid & 63;
                        uint32_t at = 0;
                        if (lane == __ffsll((long long)bal) - 1) at = atomicAdd(&sh_cnt, (uint32_t)__popcll(bal));
                        at = __shfl(at, __ffsll((long long)bal) - 1, 64);
                        if (want) {
                            const uint32_t pos = at + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                            if (pos < (uint32_t)E.cap) { cand[pos] = c; acc[t] = 0.0; }
                        }
                    }
                    if (stepwise) {
                        __syncthreads();
                        if (sh_cnt + blockDim.x > (uint32_t)E.cap) isim_cut(cand, E.K, E.cap, hist, &sh_cnt, &sh_tau, sh_aux);
                    }
                }
                __syncthreads();
                again = false;
                if (!stepwise && sh_cnt > (uint32_t)E.cap) {   // block-uniform: some candidates did not fit
                    __syncthreads();
                    if (tid == 0) sh_cnt = (uint32_t)E.cap;
                    __syncthreads();
                    isim_cut(cand, E.K, E.cap, hist, &sh_cnt, &sh_tau, sh_aux);
                    again = true;
                }
            }
        }
        isim_select(cand, (int)sh_cnt, E.K, hist, &sh_cnt, &sh_tau, sh_aux);   // then only ~K entries are left to sort
        const int n = (int)sh_cnt;
        int P2 = 1;
        while (P2 < n) P2 <<= 1;
        for (int i = n + tid; i < P2; i += blockDim.x) cand[i] = 0ull;
        __syncthreads();
        isim_sort_desc(cand, P2);
        const int keep = min(n, E.K);
        if (tid == 0) E.part_cnt[slot] = keep;
        for (int i = tid; i < keep; i += blockDim.x) E.part[(int64_t)slot * E.K + i] = cand[i];
        __syncthreads();   // everybody is done with sh_cnt / cand before the next item resets them
    }
}

// packed CSR (fy_cooc.hpp): column index relative to its chunk | the raw rating (or 1 for the co-occurrence count) as fp16
__global__ void k_isim_pack_csr(int64_t nnz, int32_t CH, const int32_t* __restrict__ csr_idx, const float* __restrict__ csr_r, int cosine,
                                uint32_t* __restrict__ pk) {
    for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < nnz; f += (int64_t)gridDim.x * blockDim.x)
        pk[f] = (uint32_t)(csr_idx[f] % CH) | ((uint32_t)__half_as_ushort(__float2half(cosine ? csr_r[f] : 1.0f)) << 16);
}
__global__ void k_isim_raw_weights(int64_t nnz, const float* __restrict__ csc_r, int cosine, float* __restrict__ csc_w) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x)
        csc_w[q] = cosine ? csc_r[q] : 1.0f;
}
__global__ void k_isim_inv_norms(int32_t Ic, const int32_t* __restrict__ rank_pair, const double* __restrict__ norm, double* __restrict__ inv) {
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < Ic; r += gridDim.x * blockDim.x) inv[r] = 1.0 / norm[rank_pair[r]];
}

constexpr int ISIM_HEAVY = 4096;   // raters above which a row is split by column chunk (test hook: FY_ISIM_HEAVY)
__global__ void k_isim_count_heavy(int32_t rows_mine, int32_t rank, int32_t world, const int32_t* __restrict__ rank_pair,
                                   const int32_t* __restrict__ pair_start, int32_t heavy, int32_t* __restrict__ n_heavy) {
    for (int32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < rows_mine; m += gridDim.x * blockDim.x) {
        const int32_t pr = rank_pair[rank + m * world];
        if (pair_start[pr + 1] - pair_start[pr] > heavy) atomicAdd(n_heavy, 1);
    }
}

// one workgroup per row: fold the chunks' top-K lists (each sorted, disjoint columns) into the row's top K
__global__ __launch_bounds__(256) void k_isim_merge(int32_t rows_mine, int32_t nch, int32_t K, const int32_t* __restrict__ part_cnt,
                                                    const uint64_t* __restrict__ part, int32_t* __restrict__ out_cnt,
                                                    int32_t* __restrict__ out_other, float* __restrict__ out_sim) {
    __shared__ uint64_t buf[2 * ISIM_MAX_K];
    const int tid = threadIdx.x;
    for (int m = blockIdx.x; m < rows_mine; m += gridDim.x) {
        int have = 0;
        for (int ch = 0; ch < nch; ch++) {
            const int n = part_cnt[(int64_t)m * nch + ch];
            if (n == 0) continue;    // block-uniform
            for (int i = tid; i < n; i += blockDim.x) buf[have + i] = part[((int64_t)m * nch + ch) * K + i];
            const int tot = have + n;
            if (have == 0) { have = n; __syncthreads(); continue; }   // a single list is already sorted
            int P2 = 1;
            while (P2 < tot) P2 <<= 1;
            for (int i = tot + tid; i < P2; i += blockDim.x) buf[i] = 0ull;
            __syncthreads();
            isim_sort_desc(buf, P2);
            have = min(tot, K);
        }
        __syncthreads();
        if (tid == 0) out_cnt[m] = have;
        for (int i = tid; i < have; i += blockDim.x) {
            const uint64_t c = buf[i];
            out_other[(int64_t)m * K + i] = 0x7FFFFFFF - (int32_t)(uint32_t)c;
            out_sim[(int64_t)m * K + i] = isim_order_unkey((uint32_t)(c >> 32));
        }
        __syncthreads();
    }
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

// ---- input preparation: minPrefsPerUser / maxPrefsPerUser (BaselinePreparePreferenceMatrixJob.java:104, 126-129; filmyou.h)
__global__ void k_pref_keys(int64_t n, const int32_t* __restrict__ user, const int32_t* __restrict__ item, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        key[t] = ((uint64_t)(uint32_t)user[t] << 32) | (uint32_t)item[t];
        val[t] = (uint32_t)t;
    }
}
__global__ void k_pref_heads(int64_t n, const uint64_t* __restrict__ key, uint32_t* __restrict__ head) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x)
        head[p] = (p == 0 || (key[p] >> 32) != (key[p - 1] >> 32)) ? 1u : 0u;
}
__global__ void k_pref_head_pos(int64_t n, const uint32_t* __restrict__ head, const uint32_t* __restrict__ useq, uint32_t* __restrict__ head_pos) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p <= n; p += (int64_t)gridDim.x * blockDim.x) {
        if (p == n) head_pos[useq[n - 1]] = (uint32_t)n;          // sentinel behind the last user
        else if (head[p]) head_pos[useq[p] - 1] = (uint32_t)p;
    }
}
__global__ void k_pref_keep(int64_t n, const uint32_t* __restrict__ useq, const uint32_t* __restrict__ head_pos, int32_t min_prefs, int32_t max_prefs,
                            uint32_t* __restrict__ keep) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t u = useq[p] - 1;
        const int64_t first = head_pos[u], deg = (int64_t)head_pos[u + 1] - first, k = p - first;
        bool ok = deg >= min_prefs;
        if (ok && max_prefs > 0 && deg > max_prefs) ok = ((k + 1) * max_prefs) / deg > (k * max_prefs) / deg;
        keep[p] = ok ? 1u : 0u;
    }
}
__global__ void k_pref_emit(int64_t n, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ off, const uint32_t* __restrict__ src,
                            const int32_t* __restrict__ user, const int32_t* __restrict__ item, const float* __restrict__ score,
                            int32_t* __restrict__ ou, int32_t* __restrict__ oi, float* __restrict__ os) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        if (!keep[p]) continue;
        const uint32_t t = src[p], o = off[p] - 1;
        ou[o] = user[t];
        oi[o] = item[t];
        os[o] = score[t];
    }
}
// a filtered copy of the ratings, or nullptr when neither option changes anything
static std::unique_ptr<fy_ratings> filter_user_prefs(Context* ctx, const fy_ratings* R, int32_t min_prefs, int32_t max_prefs) {
    if ((min_prefs <= 1 && max_prefs <= 0) || R->nnz == 0) return nullptr;
    if (R->nnz >= ((int64_t)1 << 32)) FY_FAIL(FY_ERR_UNSUPPORTED, "minPrefsPerUser / maxPrefsPerUser with more than 2^32 ratings");
    hipStream_t st = ctx->stream;
    const int64_t n = R->nnz;
    DevBuf<uint64_t> ka(ctx, n), kb(ctx, n);
    DevBuf<uint32_t> va(ctx, n), vb(ctx, n), head(ctx, n), useq(ctx, n), keep(ctx, n), off(ctx, n), head_pos(ctx, (size_t)n + 1);
    k_pref_keys<<<grid_for(n), 256, 0, st>>>(n, R->user.get(), R->item.get(), ka.get(), va.get());
    FY_KERNEL_CHECK();
    sort_pairs_u64_u32(ctx, ka.get(), kb.get(), va.get(), vb.get(), (size_t)n);
    k_pref_heads<<<grid_for(n), 256, 0, st>>>(n, kb.get(), head.get());
    FY_KERNEL_CHECK();
    inclusive_scan_u32(ctx, head.get(), useq.get(), (size_t)n);
    k_pref_head_pos<<<grid_for(n + 1), 256, 0, st>>>(n, head.get(), useq.get(), head_pos.get());
    FY_KERNEL_CHECK();
    k_pref_keep<<<grid_for(n), 256, 0, st>>>(n, useq.get(), head_pos.get(), min_prefs, max_prefs, keep.get());
    FY_KERNEL_CHECK();
    inclusive_scan_u32(ctx, keep.get(), off.get(), (size_t)n);
    const uint32_t kept = fetch(ctx, off.get() + (n - 1));
    std::unique_ptr<fy_ratings> F(new fy_ratings);
    F->ctx = ctx;
    F->nnz = kept;
    F->max_user = R->max_user;
    F->max_item = R->max_item;
    F->user.alloc(ctx, kept);
    F->item.alloc(ctx, kept);
    F->score.alloc(ctx, kept);
    k_pref_emit<<<grid_for(n), 256, 0, st>>>(n, keep.get(), off.get(), vb.get(), R->user.get(), R->item.get(), R->score.get(), F->user.get(),
                                             F->item.get(), F->score.get());
    FY_KERNEL_CHECK();
    sync(ctx);     // the scratch arrays of this function go back to the allocator
    return F;
}

fy_result* itemsim_build(Context* ctx, const fy_itemsim_params* prm, const fy_ratings* R_in) {
    if (prm->min_prefs_per_user < 0 || prm->max_prefs_per_user < 0) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "minPrefsPerUser / maxPrefsPerUser must be >= 0");
    const std::unique_ptr<fy_ratings> filtered = filter_user_prefs(ctx, R_in, prm->min_prefs_per_user, prm->max_prefs_per_user);
    const fy_ratings* R = filtered ? filtered.get() : R_in;
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
    // packed row kernel (4-byte CSR entries, norms applied in the epilogue) when every rating is fp16-exact
    const char* pk_env = getenv("FY_COOC_PK");
    const bool use_pk = P.ratings_fp16_exact && !(pk_env && atoi(pk_env) == 0);
    DevBuf<double> norm(ctx, Ic), inv_norm(ctx, Ic);
    DevBuf<float> csc_w(ctx, P.nnz), csr_w(ctx, use_pk ? 1 : (size_t)P.nnz);
    DevBuf<uint32_t> csr_pk(ctx, use_pk ? (size_t)P.nnz : 1);
    k_item_norms<<<grid_for((int64_t)Ic * 64, 256), 256, 0, st>>>(Ic, P.pair_start.get(), P.csc_r.get(), norm.get());
    FY_KERNEL_CHECK();
    if (use_pk) {
        k_isim_raw_weights<<<grid_for(P.nnz), 256, 0, st>>>(P.nnz, P.csc_r.get(), cosine, csc_w.get());
        FY_KERNEL_CHECK();
        k_isim_inv_norms<<<grid_for(Ic), 256, 0, st>>>(Ic, P.rank_pair.get(), norm.get(), inv_norm.get());
        FY_KERNEL_CHECK();
    } else {
        k_csc_weights<<<grid_for(P.nnz), 256, 0, st>>>(P.nnz, P.csc_pair.get(), P.csc_r.get(), norm.get(), cosine, csc_w.get());
        FY_KERNEL_CHECK();
        k_csr_weights<<<grid_for(P.nnz), 256, 0, st>>>(P.nnz, P.csr_idx.get(), P.csr_r.get(), P.rank_pair.get(), norm.get(), cosine, csr_w.get());
        FY_KERNEL_CHECK();
    }

    const int K = prm->max_similarities_per_item;
    // LDS = fp64 accumulators + candidate buffer <= 160 KiB.  (A 512-entry buffer would allow three column chunks instead of
    // four at ML-25M shape, but the extra cuts cost more than the shorter walk gains: 28.5 against 27.2 ms.)
    const int cap = ISIM_CAP;
    int max_ch = ((160 * 1024 - cap * 8 - 2048) / 8) / 256 * 256;
    if (const char* e = getenv("FY_COOC_MAX_CH")) { const int v = atoi(e); if (v >= 64 && v <= max_ch) max_ch = v; }   // test hook: force column chunks
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
        if (use_pk) {
            k_isim_pack_csr<<<grid_for(P.nnz), 256, 0, st>>>(P.nnz, CH, P.csr_idx.get(), P.csr_r.get(), cosine, csr_pk.get());
            FY_KERNEL_CHECK();
        }
        CoocArgs CA{P.rank_pair.get(), P.pair_start.get(), segs.ptr.get(), segs.seg.get(), segs.w.get(), P.csr_idx.get(),
                    csr_w.get(), 0, 0, Ic, CH, nch, prm->rank, rows_mine, 0, (int32_t)P.nnz, nullptr, prm->world,
                    use_pk ? csr_pk.get() : nullptr, nullptr, (uint32_t)std::min<int64_t>((int64_t)P.nnz * 4, 0xFFFFFFFFll)};
        DevBuf<int2> item_seg(ctx, (size_t)rows_mine * nch);
        DevBuf<int32_t> part_cnt(ctx, (size_t)rows_mine * nch);
        part_cnt.zero();
        // rows are in popularity order: the rows with more than ISIM_HEAVY raters are a prefix
        int32_t heavy_raters = ISIM_HEAVY;
        if (const char* e = getenv("FY_ISIM_HEAVY")) heavy_raters = std::max(0, atoi(e));
        DevBuf<int32_t> d_heavy(ctx, 1);
        d_heavy.zero();
        k_isim_count_heavy<<<grid_for(rows_mine), 256, 0, st>>>(rows_mine, prm->rank, prm->world, P.rank_pair.get(), P.pair_start.get(), heavy_raters, d_heavy.get());
        FY_KERNEL_CHECK();
        const int32_t heavy_rows = nch > 1 ? fetch(ctx, d_heavy.get()) : 0;
        const int64_t n_items = (int64_t)heavy_rows * nch + (rows_mine - heavy_rows);
        DevBuf<uint64_t> part(ctx, (size_t)rows_mine * nch * K);
        k_item_segments<<<grid_for((int64_t)rows_mine * nch), 256, 0, st>>>(CA, item_seg.get());
        FY_KERNEL_CHECK();
        CA.item_seg = item_seg.get();
        ISimEpilogue IE{P.rank_item_raw.get(), K, prm->exclude_self, prm->has_threshold, (float)prm->threshold, prm->rank,
                        prm->world, cnt.get(), other.get(), sim.get(), (use_pk && cosine) ? inv_norm.get() : nullptr,
                        part_cnt.get(), part.get(), heavy_rows, (int32_t)n_items, cap};
        const size_t lds = (size_t)CH * 8 + (size_t)cap * 8;
        FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_itemsim<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_itemsim<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int block = lds > 48 * 1024 ? 1024 : 256;
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2048 / block, (160 * 1024) / (lds + 1024)));
        const int grid = (int)std::min<int64_t>(n_items, (int64_t)ctx->num_cus * per_cu);
        DevBuf<int32_t> next_row(ctx, 1);
        next_row.zero();
        const size_t sp = t_cooc.begin();
        if (use_pk) k_cooc_itemsim<true><<<grid, block, lds, st>>>(CA, IE, next_row.get());
        else k_cooc_itemsim<false><<<grid, block, lds, st>>>(CA, IE, next_row.get());
        FY_KERNEL_CHECK();
        k_isim_merge<<<std::min(rows_mine, ctx->num_cus * 8), 256, 0, st>>>(rows_mine, nch, K, part_cnt.get(), part.get(), cnt.get(), other.get(),
                                                                            sim.get());
        FY_KERNEL_CHECK();
        t_cooc.end(sp);
        sync(ctx);   // part / item_seg go back to the allocator at the end of this scope
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
