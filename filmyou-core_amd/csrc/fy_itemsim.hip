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
// Two builds:
//  * symmetric (round 3; cosine on positive fp16-exact ratings, one rank, the benchmark sizes): the product is symmetric, so
//    only its upper triangle is accumulated -- by the RM2 row kernel itself (fy_rm2.hip: gram_half_build: symmetric walk,
//    fixed-point ds_add_u64, half the pair visits) into an fp32 matrix in HBM -- and k_isim_sweep streams every element ONCE
//    for BOTH rows it belongs to: G[i][j] (i < j) is a candidate of row i where it is read along the row and of row j where
//    it is read down the column, 64 rows (= 256-byte row segments) at a time.  No mirror pass.
//  * row at a time (every other configuration): a workgroup owns item row i, accumulates it chunk by chunk in LDS (fp64) and
//    keeps a running top-K in LDS; no I x I matrix is written, only I x K rows leave the chip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

#include "fy_cooc.hpp"
#include "fy_prep.hpp"
#include "fy_rm2.hpp"

namespace fy {

static inline int grid_for(int64_t n, int block = 256, int cap = 256 * 16) {
    int64_t g = ceil_div(n, block);
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

// one WORKGROUP per item (pair): L2 norm of the item's rating column in a fixed summation order (thread-strided partial sums, wave
// shuffles, the four waves' partials in order), and -- round 4 -- the column's rating sum and largest rating in the same pass (the
// bounds of the fixed-point scale of the symmetric build; k_isim_gram_prep walked the columns a second time for them).  A wave per
// item made both passes wait for the heaviest column: 81 k entries on one wave = 1 270 dependent trips, 0.6 ms each pass.
__global__ __launch_bounds__(256) void k_item_norms(int32_t nP, const int32_t* __restrict__ pair_start, const float* __restrict__ csc_r,
                                                    double* __restrict__ norm, float* __restrict__ colsum, float* __restrict__ colmax) {
    __shared__ double sh_s[4];
    __shared__ float sh_sum[4], sh_max[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int32_t p = blockIdx.x; p < nP; p += gridDim.x) {
        double s = 0.0;
        float sum = 0.0f, mx = 0.0f;
        for (int32_t q = pair_start[p] + threadIdx.x; q < pair_start[p + 1]; q += 256) {
            const float r = csc_r[q];
            s += (double)r * (double)r;
            sum += r;
            mx = fmaxf(mx, r);
        }
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_down(s, o, 64);
            sum += __shfl_down(sum, o, 64);
            mx = fmaxf(mx, __shfl_down(mx, o, 64));
        }
        if (lane == 0) { sh_s[wave] = s; sh_sum[wave] = sum; sh_max[wave] = mx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            norm[p] = sqrt(((sh_s[0] + sh_s[1]) + sh_s[2]) + sh_s[3]);
            if (colsum) colsum[p] = ((sh_sum[0] + sh_sum[1]) + sh_sum[2]) + sh_sum[3];
            if (colmax) colmax[p] = fmaxf(fmaxf(sh_max[0], sh_max[1]), fmaxf(sh_max[2], sh_max[3]));
        }
        __syncthreads();
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


// ================================================================ symmetric build: band sweep over the fp32 half Gram
// G (fy_rm2.hip: gram_half_build): row i holds sum_v r_vi r_vj for the columns j > i (exact zeros for (i & ~255) <= j <= i,
// nothing in front).  sim(i, j) = G * inv_norm[i] * inv_norm[j].  A workgroup owns a PIECE = (band of 64 rows [i0, i0 + 64)) x
// (column range [c0, c1)):
//   * "down the column": for every row j in [c0, c1) above the band (j < i0 + lane), the 256-byte segment G[j][i0 .. i0 + 64) --
//     lane l looks at the candidate j of ITS row i0 + l;
//   * "along the row": for every row of the band, G[i][max(c0, i0) .. c1) in 1 KB tiles -- all lanes look at candidates of one row.
// Every element of the triangle is read exactly once by each of the two rows it belongs to; a band's pieces have equal size.
// Running top-K without a sort: a candidate must reach the row's threshold key tau (float bits order like the floats: all
// similarities here are > 0).  Passing candidates are appended to the row's list in HBM and counted in the row's 64-bucket
// histogram of their keys (8 buckets per octave below 1.0; in HBM / L2, shared by the band's pieces); every 8th candidate of a
// row that has K, tau rises to the lower edge of the highest bucket with K candidates at or above it -- a valid lower bound of
// the row's final K-th best (they are K real candidates of the row), published to the band's other pieces through tau_g.  After warm-up a few candidates per thousand
// pass; k_isim_finish selects and sorts the lists.  A row whose list overflows (massive ties inside one bucket) is redone
// exactly by k_isim_finish from the matrix.
constexpr int SWEEP_KEY_SHIFT = 20, SWEEP_KEY_BASE = 1016 - 63;   // bucket = (key >> 20) - base: 63 <-> [1, 1.125), 0 <-> everything below 2^-7.875
struct SweepArgs {
    const float* __restrict__ G;
    int64_t ldm;
    int32_t Ic, K;
    const float* __restrict__ invn32;          // [ldm + slack] 1 / norm in rank order, fp32 (0 behind Ic): the prefilter
    const double* __restrict__ invn;           // [Ic]
    const int32_t* __restrict__ rank_item_raw;
    uint32_t* __restrict__ tau_g;              // [Ic]: key a candidate of the row must reach
    int32_t* __restrict__ gcnt;                // [Ic]: candidates appended
    uint64_t* __restrict__ glist;              // [Ic][capg]: key << 32 | (0x7FFFFFFF - raw item id)
    int32_t capg;
    int32_t* __restrict__ overflow;            // [Ic]
    uint32_t* __restrict__ hist_g;             // [Ic][64]: the row's candidates by key bucket (all pieces of the band add to it)
    int32_t nbands, piece;
    uint32_t tau0;                             // the job's own threshold as a key (1 = "similarity > 0")
    int32_t exclude_self;
    int32_t* __restrict__ out_cnt;             // k_isim_finish: [Ic], [Ic * K], [Ic * K]
    int32_t* __restrict__ out_other;
    float* __restrict__ out_sim;
};

// wave-level: lanes with `pass` hold a candidate (row i = band row rl, column col, raw Gram value v) that survived the fp32
// prefilter; exact value, append, histogram, and -- every 8th candidate of a row once it has K -- a new threshold for the row
__device__ __forceinline__ void sweep_take(const SweepArgs& A, uint32_t* tau_l, bool pass, int rl, int i, int col, float v, int i0) {
    const int lane = threadIdx.x & 63;
    bool trig = false;
    if (pass) {
        const float sf = (float)((double)v * A.invn[i] * A.invn[col]);
        const uint32_t key = __float_as_uint(sf);
        if (sf > 0.0f && key >= tau_l[rl]) {
            int bk = (int)(key >> SWEEP_KEY_SHIFT) - SWEEP_KEY_BASE;
            bk = bk < 0 ? 0 : (bk > 63 ? 63 : bk);
            atomicAdd(&A.hist_g[(int64_t)i * 64 + bk], 1u);
            const int pos = atomicAdd(&A.gcnt[i], 1);
            if (pos < A.capg) A.glist[(int64_t)i * A.capg + pos] = ((uint64_t)key << 32) | (uint32_t)(0x7FFFFFFF - A.rank_item_raw[col]);
            else A.overflow[i] = 1;
            trig = pos + 1 >= A.K && ((pos + 1) & 7) == 0;
        }
    }
    unsigned long long m = __ballot(trig);
    while (m) {   // wave-uniform
        const int src = __ffsll((long long)m) - 1;
        const int r = __shfl(rl, src, 64);
        m &= ~__ballot(rl == r);                       // (along a row every lane holds the same row)
        // suffix sums over the row's buckets: candidates at or above bucket `lane`, over ALL pieces of the band (the histogram
        // is the row's, in HBM / L2: the first version kept one per piece in LDS, and a piece sees a seventh of the row -- its K-th
        // best is the row's 7 K-th: 1157 candidates per row for K = 100)
        uint32_t h = A.hist_g[(int64_t)(i0 + r) * 64 + lane];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_down((int)h, o, 64);
            if (lane + o < 64) h += t;
        }
        const unsigned long long ok = __ballot(h >= (uint32_t)A.K);
        if (ok) {
            const int B = 63 - __clzll((long long)ok);
            if (B > 0 && lane == 0) {
                const uint32_t nt = (uint32_t)(B + SWEEP_KEY_BASE) << SWEEP_KEY_SHIFT;
                atomicMax(&tau_l[r], nt);
                atomicMax(&A.tau_g[i0 + r], nt);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_isim_sweep(SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) float fy_sweep_lds[];      // [piece + 256]: 1 / norm of the piece's columns, then 64: of the band's rows
    __shared__ uint32_t tau_l[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = blockIdx.x / A.nbands, b = blockIdx.x - p * A.nbands;
    const int i0 = b * 64;
    const int c0 = p * A.piece, c1 = min(c0 + A.piece, (int)A.ldm);
    float* __restrict__ nrm_cols = fy_sweep_lds;
    float* __restrict__ nrm_rows = fy_sweep_lds + A.piece + 256;
    // the norms the piece needs, once: every element used to bring its column's norm along from L2 (a second 16-byte load per
    // 16 bytes of the matrix), and every row of the band began with a dependent load of its own norm
    for (int t = tid; t < A.piece + 256; t += 256) nrm_cols[t] = c0 + t < (int)A.ldm ? A.invn32[c0 + t] : 0.0f;
    if (tid < 64) {
        nrm_rows[tid] = A.invn32[i0 + tid];
        tau_l[tid] = i0 + tid < A.Ic ? A.tau_g[i0 + tid] : 0xFFFFFFFFu;
    }
    __syncthreads();
    // The fast path of both parts is straight-line: U loads in flight, one compare per element against the row's threshold in
    // fp32 with a margin, the outcome kept as one bit per element.  Only when some lane of the wave holds a set bit does the
    // wave enter the slow path -- ONE copy of sweep_take per part, each lane presenting its own lowest set element.  (The first
    // version called sweep_take from every unrolled compare: 24 inlined copies, and it read the column-side norms with
    // wave-uniform loads that the compiler turned into dependent vector loads behind s_waitcnt vmcnt(0): 8.6 ms for 14 GB.)
    // ---- down the column: rows j of the matrix in [c0, jend), segment [i0, i0 + 64); wave w takes j = c0 + w, c0 + w + 4, ...
    const int jend = min(min(c1, i0 + 64), A.Ic);
    if (c0 < jend && A.exclude_self != 3) {
        const int i = i0 + lane;
        const bool valid = i < A.Ic;
        const float my_inv = valid ? nrm_rows[lane] : 0.0f;
        const float* __restrict__ gp = A.G + i0 + lane;
        constexpr int U = 8;
        for (int jb = c0 + wave; jb < jend; jb += 4 * 64) {     // 64 of the wave's rows per block: their norms in one read, lane k <-> row jb + 4 k
            const float nj_vec = nrm_cols[min(jb + 4 * lane, jend - 1) - c0];
            for (int k0 = 0; k0 < 64 && jb + 4 * k0 < jend; k0 += U) {     // wave-uniform
                float v[U];
#pragma unroll
                for (int u = 0; u < U; u++) v[u] = gp[(int64_t)min(jb + 4 * (k0 + u), jend - 1) * A.ldm];     // unconditional loads (a clamped row is masked below)
                const float tauf = __uint_as_float(tau_l[lane]) * 0.999999f;
                uint32_t pm = 0;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int jj = jb + 4 * (k0 + u);
                    const float nj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(nj_vec), k0 + u));
                    const float s = v[u] * nj * my_inv;
                    // (bitwise, not &&: short-circuit evaluation turned these into exec-masked branches around the loads)
                    pm |= ((uint32_t)valid & (uint32_t)(jj < jend) & (uint32_t)(jj < i) & (uint32_t)(s > 0.0f) & (uint32_t)(s >= tauf)) << u;
                }
                while (__ballot(pm != 0)) {      // slow path: rare once the thresholds have warmed up
                    const int u = pm ? __ffs((int)pm) - 1 : 0;
                    float vu = v[0];
#pragma unroll
                    for (int q = 1; q < U; q++) vu = u == q ? v[q] : vu;
                    sweep_take(A, tau_l, pm != 0, lane, i, jb + 4 * (k0 + u), vu, i0);
                    pm &= pm - 1;
                }
            }
        }
    }
    // ---- along the row: columns [cs, c1) of the band's rows (a row's candidates are the columns behind it).  ONE loop over the
    // (row, 256-column tile) pairs of this wave -- rows wave, wave + 4, ... -- four tiles in flight, also across the end of a row
    const int cs = max(c0, i0);
    const int n_rows = min(64, A.Ic - i0);
    if (cs < c1 && wave < n_rows && A.exclude_self != 2) {
        constexpr int U = 4;
        const int tiles_per_row = (c1 - cs + 255) >> 8;
        const int total = ((n_rows - wave + 3) >> 2) * tiles_per_row;
        int rr = wave, tt = 0;                                  // row / tile of pair T0 (wave-uniform)
        for (int T0 = 0; T0 < total; T0 += U) {
            float4 v[U];
            int r_of[U], col_of[U];
            {
                int r2 = rr, t2 = tt;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    r_of[u] = r2;
                    col_of[u] = cs + 256 * t2 + 4 * lane;      // (a tile that overshoots c1 reads the slack behind the row / the matrix: masked below)
                    v[u] = *reinterpret_cast<const float4*>(A.G + (int64_t)(i0 + r2) * A.ldm + col_of[u]);
                    if (T0 + u + 1 < total) { if (++t2 == tiles_per_row) { t2 = 0; r2 += 4; } }     // (the last pairs repeat: masked by T0 + u < total)
                }
                rr = r2; tt = t2;                             // = pair T0 + U
            }
            float vv[4 * U];
            uint32_t pm = 0;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + r_of[u];
                const float ri = nrm_rows[r_of[u]];
                const float tauf = __uint_as_float(tau_l[r_of[u]]) * 0.999999f;
                const float4 n4 = *reinterpret_cast<const float4*>(nrm_cols + (col_of[u] - c0));
                vv[4 * u + 0] = v[u].x; vv[4 * u + 1] = v[u].y; vv[4 * u + 2] = v[u].z; vv[4 * u + 3] = v[u].w;
                const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int col = col_of[u] + e;
                    const float s = vv[4 * u + e] * ri * nn[e];
                    pm |= ((uint32_t)(T0 + u < total) & (uint32_t)(col < c1) & (uint32_t)(col > i) & (uint32_t)(col < A.Ic) & (uint32_t)(s > 0.0f) & (uint32_t)(s >= tauf)) << (4 * u + e);
                }
            }
            while (__ballot(pm != 0)) {
                const int x = pm ? __ffs((int)pm) - 1 : 0;
                float vx = vv[0];
#pragma unroll
                for (int q = 1; q < 4 * U; q++) vx = x == q ? vv[q] : vx;
                int rx = r_of[0], cx = col_of[0];
#pragma unroll
                for (int q = 1; q < U; q++) { rx = (x >> 2) == q ? r_of[q] : rx; cx = (x >> 2) == q ? col_of[q] : cx; }
                sweep_take(A, tau_l, pm != 0, rx, i0 + rx, cx + (x & 3), vx, i0);
                pm &= pm - 1;
            }
        }
    }
}

// (LDS traffic between the lanes of ONE wave needs no s_barrier -- a wave's LDS instructions complete in order -- only a
// compiler fence, so that a lane's read of another lane's word is not moved in front of the writes)
__device__ __forceinline__ void isim_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// the diagonal candidate of row i (not part of the triangle): sum r^2 / norm^2; 0 = none
__device__ __forceinline__ uint64_t isim_self_candidate(const SweepArgs& A, int i) {
    if (A.exclude_self) return 0ull;
    const double inv_i = A.invn[i], nrm = 1.0 / inv_i;
    const float sf = (float)(nrm * nrm * inv_i * inv_i);
    const uint32_t key = __float_as_uint(sf);
    return (sf > 0.0f && key >= A.tau0) ? (((uint64_t)key << 32) | (uint32_t)(0x7FFFFFFF - A.rank_item_raw[i])) : 0ull;
}

// One WAVE per row: the K best of the row's candidate list (a few hundred entries of 8 bytes, L2-resident), sorted by
// descending similarity, ties by ascending item id -- the order of the composite key << 32 | (0x7FFFFFFF - id), all distinct.
// More than FIN_SMALL candidates are first cut by a radix select over the composite, 8 bits per level from the top, until the
// entries at or above the prefix fit the sort buffer (two levels = 16 key bits in practice; eight levels are exact for any
// input).  No workgroup barrier anywhere: 59 047 workgroups with ~40 barriers each were 2.0 ms, this is one pass of waves.
// Rows whose list overflowed in the sweep are handed to k_isim_redo.
constexpr int FIN_SURV = 512, FIN_SMALL = 128;
__global__ __launch_bounds__(256) void k_isim_finish(SweepArgs A, int32_t* __restrict__ redo_list, int32_t* __restrict__ n_redo) {
    __shared__ uint64_t surv_all[4][FIN_SURV];
    __shared__ uint32_t hist_all[4][256];
    __shared__ uint32_t scal_all[4][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + w;
    if (i >= A.Ic) return;                               // wave-uniform; nothing below synchronises the workgroup
    if (A.overflow[i]) {
        if (lane == 0) redo_list[atomicAdd(n_redo, 1)] = i;
        return;
    }
    uint64_t* surv = surv_all[w];
    uint32_t* hist = hist_all[w];
    uint32_t* scal = scal_all[w];
    const int n = min(A.gcnt[i], A.capg);
    const uint64_t* __restrict__ L = A.glist + (int64_t)i * A.capg;
    const uint64_t c_self = isim_self_candidate(A, i);
    const int m = n + (c_self ? 1 : 0);
    auto get = [&](int t) -> uint64_t { return t < n ? L[t] : c_self; };
    int cnt = 0;
    if (m <= FIN_SMALL || m <= A.K) {                   // (K <= 512 = FIN_SURV here: itemsim_symmetric)
        for (int t = lane; t < m && t < FIN_SURV; t += 64) surv[t] = get(t);
        cnt = min(m, FIN_SURV);
        if (m > FIN_SURV) cnt = -1;                      // K > FIN_SURV never reaches this kernel
    } else {
        uint64_t prefix = 0;
        uint32_t above = 0;
        int shift = 56;
        for (;; shift -= 8) {
            for (int bb = lane; bb < 256; bb += 64) hist[bb] = 0u;
            isim_wave_sync();
            const uint64_t hi_mask = shift == 56 ? 0ull : ~0ull << (shift + 8);
            for (int t = lane; t < m; t += 64) {
                const uint64_t c = get(t);
                if ((c & hi_mask) == prefix) atomicAdd(&hist[(uint32_t)(c >> shift) & 255u], 1u);
            }
            isim_wave_sync();
            if (lane == 0) {
                uint32_t cum = above;
                int bb = 255;
                for (; bb > 0; bb--) {
                    if (cum + hist[bb] >= (uint32_t)A.K) break;
                    cum += hist[bb];
                }
                scal[0] = (uint32_t)bb;
                scal[1] = cum;
                scal[2] = cum + hist[bb];
            }
            isim_wave_sync();
            prefix |= (uint64_t)scal[0] << shift;
            above = scal[1];
            const uint32_t at_or_above = scal[2];
            isim_wave_sync();
            if (at_or_above <= (uint32_t)FIN_SURV || shift == 0) break;      // (shift == 0: the prefix is the K-th composite itself)
        }
        if (lane == 0) scal[3] = 0u;
        isim_wave_sync();
        for (int t0 = 0; t0 < m; t0 += 64) {
            const int t = t0 + lane;
            const uint64_t c = t < m ? get(t) : 0ull;
            const bool keep = t < m && c >= prefix;
            const unsigned long long bal = __ballot(keep);
            if (bal) {
                const uint32_t base = scal[3];
                if (keep) {
                    const uint32_t pos = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    if (pos < (uint32_t)FIN_SURV) surv[pos] = c;
                }
                isim_wave_sync();
                if (lane == 0) scal[3] = base + (uint32_t)__popcll(bal);
                isim_wave_sync();
            }
        }
        cnt = (int)min(scal[3], (uint32_t)FIN_SURV);
    }
    int P2 = 1;
    while (P2 < cnt) P2 <<= 1;
    for (int t = cnt + lane; t < P2; t += 64) surv[t] = 0ull;
    isim_wave_sync();
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < P2; t += 64) {
                const int l = t ^ j;
                if (l > t) {
                    const uint64_t x = surv[t], y = surv[l];
                    const bool desc = (t & k) == 0;
                    if (desc ? (x < y) : (x > y)) { surv[t] = y; surv[l] = x; }
                }
            }
            isim_wave_sync();
        }
    const int keep = min(cnt, A.K);
    if (lane == 0) A.out_cnt[i] = keep;
    for (int t = lane; t < keep; t += 64) {
        const uint64_t c = surv[t];
        A.out_other[(int64_t)i * A.K + t] = 0x7FFFFFFF - (int32_t)(uint32_t)c;
        A.out_sim[(int64_t)i * A.K + t] = __uint_as_float((uint32_t)(c >> 32));
    }
}

// rows whose candidate list overflowed (massive ties inside one bucket of the sweep's histogram): every column of the row, from
// the matrix, through the running top-K of the row-at-a-time kernel (isim_cut: exact for any input)
__global__ __launch_bounds__(256) void k_isim_redo(SweepArgs A, const int32_t* __restrict__ redo_list, const int32_t* __restrict__ n_redo) {
    __shared__ uint64_t cand[ISIM_CAP];
    __shared__ uint32_t sh_cnt, sh_tau, sh_aux[2], hist[256];
    const int tid = threadIdx.x;
    const int n_rows = *n_redo;
    for (int x = blockIdx.x; x < n_rows; x += gridDim.x) {      // block-uniform
        const int i = redo_list[x];
        const double inv_i = A.invn[i];
        if (tid == 0) { sh_cnt = 0; sh_tau = A.tau0; }
        __syncthreads();
        for (int base = 0; base < A.Ic; base += 256) {
            const int col = base + tid;
            bool want = false;
            uint64_t c = 0;
            if (col < A.Ic && col != i) {
                const float v = col > i ? A.G[(int64_t)i * A.ldm + col] : A.G[(int64_t)col * A.ldm + i];
                const float sf = (float)((double)v * inv_i * A.invn[col]);
                const uint32_t key = __float_as_uint(sf);
                want = sf > 0.0f && key >= sh_tau;
                c = ((uint64_t)key << 32) | (uint32_t)(0x7FFFFFFF - A.rank_item_raw[col]);
            }
            const unsigned long long bal = __ballot(want);
            if (bal) {
                const int lane = tid & 63;
                uint32_t at = 0;
                if (lane == __ffsll((long long)bal) - 1) at = atomicAdd(&sh_cnt, (uint32_t)__popcll(bal));
                at = __shfl(at, __ffsll((long long)bal) - 1, 64);
                if (want) cand[at + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = c;
            }
            __syncthreads();
            if (sh_cnt + 256 + 1 > (uint32_t)ISIM_CAP) isim_cut(cand, A.K, ISIM_CAP, hist, &sh_cnt, &sh_tau, sh_aux);   // block-uniform (+ 1: the diagonal)
        }
        const uint64_t c_self = isim_self_candidate(A, i);
        if (c_self && tid == 0) cand[sh_cnt++] = c_self;
        __syncthreads();
        isim_select(cand, (int)sh_cnt, A.K, hist, &sh_cnt, &sh_tau, sh_aux);
        const int n = (int)sh_cnt;
        int P2 = 1;
        while (P2 < n) P2 <<= 1;
        for (int t = n + tid; t < P2; t += 256) cand[t] = 0ull;
        __syncthreads();
        isim_sort_desc(cand, P2);
        const int keep = min(n, A.K);
        if (tid == 0) A.out_cnt[i] = keep;
        for (int t = tid; t < keep; t += 256) {
            const uint64_t c = cand[t];
            A.out_other[(int64_t)i * A.K + t] = 0x7FFFFFFF - (int32_t)(uint32_t)c;
            A.out_sim[(int64_t)i * A.K + t] = __uint_as_float((uint32_t)(c >> 32));
        }
        __syncthreads();     // everybody is done with cand / sh_cnt before the next row resets them
    }
}
__global__ void k_sum_i32(int32_t n, const int32_t* __restrict__ v, unsigned long long* __restrict__ out) {
    unsigned long long s = 0;
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) s += (unsigned long long)v[t];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// per item in rank order: 1 / norm (fp64 and fp32) and the bounds of the fixed-point scale (largest column sum / largest rating, from
// k_item_norms' per-column sums and maxima: no second walk over the columns)
__global__ void k_isim_gram_prep(int32_t Ic, int64_t ld_pad, const int32_t* __restrict__ rank_pair, const float* __restrict__ colsum,
                                 const float* __restrict__ colmax, const double* __restrict__ norm, double* __restrict__ invn, float* __restrict__ invn32,
                                 uint32_t* __restrict__ bounds /* [0] max column sum, [1] max rating, as float bits */) {
    float bs = 0.0f, bm = 0.0f;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < ld_pad; r += (int64_t)gridDim.x * blockDim.x) {
        if (r >= Ic) { invn32[r] = 0.0f; continue; }
        const int32_t pr = rank_pair[r];
        const double inv = 1.0 / norm[pr];
        invn[r] = inv;
        invn32[r] = (float)inv;
        bs = fmaxf(bs, colsum[pr] * 1.0001f);
        bm = fmaxf(bm, colmax[pr]);
    }
    // positive floats order like their bits; one atomic per wave (59 047 atomics on ONE address took 1.4 ms: 82 M/s)
    for (int o = 32; o > 0; o >>= 1) { bs = fmaxf(bs, __shfl_down(bs, o, 64)); bm = fmaxf(bm, __shfl_down(bm, o, 64)); }
    if ((threadIdx.x & 63) == 0) {
        if (__float_as_uint(bs) > bounds[0]) atomicMax(&bounds[0], __float_as_uint(bs));
        if (__float_as_uint(bm) > bounds[1]) atomicMax(&bounds[1], __float_as_uint(bm));
    }
}
__global__ void k_fill_u32(int64_t n, uint32_t v, uint32_t* __restrict__ out) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) out[t] = v;
}

// the symmetric build; false = not applicable here (the caller runs the row-at-a-time build).  cnt / other / sim as the other build leaves them.
static bool itemsim_symmetric(Context* ctx, const fy_itemsim_params* prm, const Prepared& P, const double* norm, const float* colsum, const float* colmax,
                              int32_t* cnt, int32_t* other, float* sim, double* ms_cooc, fy_stats* st_out) {
    const Tuning& tune = ctx->tune;
    const int32_t Ic = P.nP, K = prm->max_similarities_per_item;
    if (tune.isim_gram == 0 || prm->world != 1 || prm->similarity != FY_SIMILARITY_COSINE || !tune.cooc_pk || !P.ratings_fp16_exact ||
        !P.ratings_positive || K > FIN_SURV / 2 || (prm->has_threshold && !(prm->threshold > 0.0)))
        return false;
    if (tune.isim_gram < 0 && Ic < tune.isim_gram_min_items) return false;
    const int64_t ldm = round_up(Ic, 256), slack = 1024;
    if ((uint64_t)Ic * (uint64_t)ldm * 4 > ctx->total_mem / 3) return false;     // the fp32 matrix must fit comfortably
    hipStream_t st = ctx->stream;
    DevBuf<double> invn(ctx, (size_t)Ic);
    DevBuf<float> invn32(ctx, (size_t)(ldm + slack));
    DevBuf<uint32_t> d_bounds(ctx, 2);
    d_bounds.zero();
    k_isim_gram_prep<<<grid_for(ldm + slack, 256, 256), 256, 0, st>>>(Ic, ldm + slack, P.rank_pair.get(), colsum, colmax, norm, invn.get(), invn32.get(), d_bounds.get());
    FY_KERNEL_CHECK();
    uint32_t hb[2];
    d2h(ctx, hb, d_bounds.get(), 2);
    sync(ctx);
    float fsum, fmax;
    memcpy(&fsum, &hb[0], 4);
    memcpy(&fmax, &hb[1], 4);
    const float bounds3[3] = {fsum, fmax, fmax};
    DevBuf<float> G(ctx, (size_t)Ic * ldm + slack);
    EventTimer t_all(ctx);
    const size_t sp = t_all.begin();
    double ms_tables = 0.0;
    if (!gram_half_build(ctx, P, P.csc_r.get(), bounds3, G.get(), ldm, &ms_tables, nullptr)) return false;
    const int capg = tune.isim_capg;
    DevBuf<uint32_t> tau_g(ctx, (size_t)Ic);
    DevBuf<int32_t> gcnt(ctx, (size_t)Ic), overflow(ctx, (size_t)Ic);
    DevBuf<uint64_t> glist(ctx, (size_t)Ic * capg);
    gcnt.zero();
    overflow.zero();
    uint32_t tau0 = 1u;      // no threshold: similarity > 0 (NO_THRESHOLD = Double.MIN_VALUE)
    if (prm->has_threshold) {
        float tf = (float)prm->threshold;
        if ((double)tf < prm->threshold) tf = nextafterf(tf, INFINITY);      // float s >= double threshold  <=>  s >= the next float at or above it
        memcpy(&tau0, &tf, 4);
    }
    k_fill_u32<<<grid_for(Ic), 256, 0, st>>>(Ic, tau0, tau_g.get());
    FY_KERNEL_CHECK();
    DevBuf<uint32_t> hist_g(ctx, (size_t)round_up(Ic, 64) * 64);
    hist_g.zero();
    SweepArgs SA{G.get(), ldm, Ic, K, invn32.get(), invn.get(), P.rank_item_raw.get(), tau_g.get(), gcnt.get(), glist.get(), capg, overflow.get(),
                 hist_g.get(), (int32_t)ceil_div(Ic, 64), tune.isim_piece, tau0, prm->exclude_self, cnt, other, sim};
    const int npieces = (int)ceil_div(Ic, SA.piece);
    k_isim_sweep<<<SA.nbands * npieces, 256, (size_t)(SA.piece + 256 + 64) * sizeof(float), st>>>(SA);
    FY_KERNEL_CHECK();
    DevBuf<int32_t> redo_list(ctx, (size_t)Ic), n_redo(ctx, 1);
    n_redo.zero();
    k_isim_finish<<<(int)ceil_div(Ic, 4), 256, 0, st>>>(SA, redo_list.get(), n_redo.get());
    FY_KERNEL_CHECK();
    k_isim_redo<<<ctx->num_cus * 2, 256, 0, st>>>(SA, redo_list.get(), n_redo.get());
    FY_KERNEL_CHECK();
    t_all.end(sp);
    DevBuf<unsigned long long> d_cands(ctx, 1);
    d_cands.zero();
    k_sum_i32<<<grid_for(Ic), 256, 0, st>>>(Ic, gcnt.get(), d_cands.get());
    FY_KERNEL_CHECK();
    unsigned long long h_cands = 0;
    int32_t h_redo = 0;
    d2h(ctx, &h_cands, d_cands.get(), 1);
    d2h(ctx, &h_redo, n_redo.get(), 1);
    sync(ctx);
    *ms_cooc = t_all.total_ms() - ms_tables;
    st_out->ms_tables = ms_tables;
    st_out->isim_candidates = (int64_t)h_cands;
    st_out->isim_redone_rows = h_redo;
    return true;
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
    F->scores_fp16_exact = R->scores_fp16_exact;
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
    const Tuning& tune = ctx->tune;
    const bool use_pk = P.ratings_fp16_exact && tune.cooc_pk;
    DevBuf<double> norm(ctx, Ic), inv_norm(ctx, Ic);
    DevBuf<float> colsum(ctx, Ic), colmax(ctx, Ic);
    k_item_norms<<<std::min<int>(Ic, ctx->num_cus * 32), 256, 0, st>>>(Ic, P.pair_start.get(), P.csc_r.get(), norm.get(), colsum.get(), colmax.get());
    FY_KERNEL_CHECK();
    const int K = prm->max_similarities_per_item;
    const int32_t rows_mine = (Ic - prm->rank + prm->world - 1) / prm->world;
    DevBuf<int32_t> cnt(ctx, (size_t)rows_mine + 1), off(ctx, (size_t)rows_mine + 1), other(ctx, (size_t)rows_mine * K);
    DevBuf<float> sim(ctx, (size_t)rows_mine * K);
    cnt.zero();
    double ms_sym = 0.0;
    const bool symmetric = itemsim_symmetric(ctx, prm, P, norm.get(), colsum.get(), colmax.get(), cnt.get(), other.get(), sim.get(), &ms_sym, &Rs->st);
    if (symmetric) Rs->st.cooc_launches = 1;
    DevBuf<float> csc_w(ctx, symmetric ? 1 : (size_t)P.nnz), csr_w(ctx, (use_pk || symmetric) ? 1 : (size_t)P.nnz);
    DevBuf<uint32_t> csr_pk(ctx, (use_pk && !symmetric) ? (size_t)P.nnz : 1);
    if (symmetric) {
    } else if (use_pk) {
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

    // LDS = fp64 accumulators + candidate buffer <= 160 KiB.  (A 512-entry buffer would allow three column chunks instead of
    // four at ML-25M shape, but the extra cuts cost more than the shorter walk gains: 28.5 against 27.2 ms.)
    const int cap = ISIM_CAP;
    int max_ch = ((160 * 1024 - cap * 8 - 2048) / 8) / 256 * 256;
    if (tune.cooc_max_ch_forced && tune.cooc_max_ch <= max_ch) max_ch = tune.cooc_max_ch;   // test hook: force column chunks
    int32_t CH, nch;
    pick_chunks(Ic, max_ch, CH, nch);
    DevBuf<int32_t> chunk_off(ctx, (size_t)P.nU * (nch + 1));
    build_chunk_offsets(ctx, P.rowptr.get(), P.csr_idx.get(), 0, P.nU, CH, nch, chunk_off.get());
    if (rows_mine > 0 && !symmetric) {
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
        const int32_t heavy_raters = tune.isim_heavy;
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
    Rs->st.ms_cooc = symmetric ? ms_sym : t_cooc.total_ms();
    Rs->st.ms_total = t_total.total_ms();
    return Rs.release();
}

}  // namespace fy
