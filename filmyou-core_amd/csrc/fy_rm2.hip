// fy_rm2.hip -- the RM2 relevance-model job on MI355X: statistics, per-cluster M matrix, p(i|u) scoring, top-N.
//
// Math (SURVEY.md section 8a "RM2 in one place"; reference M/rm/AbstractRM2Reducer.java:185-190, 321-371, 384-389):
//     c_vi       = (1-l) r_vi / s_v + l p_i                                  (probItemGivenUser, :384-389)
//     score(u,i) = (n-1) ln M - n ln U_c + sum_{j in rated(u)} ln( sum_{v != u} c_vi c_vj )   for i unrated by u
// The reference evaluates the inner sum with U_c - 1 multiply-adds per (i, j).  Here it is hoisted: for i NOT rated by u
// (x_ui = 0, x = r/s)
//     sum_{v != u} c_vi c_vj = M[j][i] + (l p_i) * e_uj
//     M[j][i] = (1-l)^2 (X^T X)_ij + l (1-l) p_j b_i          per cluster, dense fp32 in HBM, b_i = sum_v x_vi
//     e_uj    = (1-l) (b_j - x_uj) + l (U_c - 1) p_j           per rating of u (= sum_{v != u} c_vj), fp32
// Every term of that form is non-negative, so there is no cancellation even for 2-user clusters; U_c = 1 gives
// exactly 0 -> ln 0 = -inf like the reference (quirk Q7).  Logs are v_log_f32 (base 2, scaled by ln 2 at the end);
// eight of them are added in fp32 and folded into an fp64 running sum, which keeps the relative error of a
// 10^4-term score near 1e-7 (tolerance of north_star: 1e-5).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "fy_cooc.hpp"
#include "fy_prep.hpp"
#include <chrono>
#include "fy_rm2.hpp"

namespace fy {

static inline int grid_for(int64_t n, int block = 256, int cap = 256 * 16) {
    int64_t g = ceil_div(n, block);
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

// ================================================================ statistics (jobs RM2-1 / RM2-2)
// One walk over the CSC gives everything that is summed per (cluster, item) column: this rank's PARTIAL rating sum (users
// in slots [lo, hi), the exchange buffer), b = sum_v r_vi / s_v, and the row kernel's work model sum_v n_v.
// Columns are processed in popularity-rank order.  A wave owns a column of up to PAIR_HEAVY raters; the few heavier
// columns (the most popular item of ML-25M shape has ~10^5 raters: one wave would walk it for a millisecond) are
// appended to a list and taken by whole workgroups in k_pair_pass_heavy.  Every column is reduced in a fixed order.
constexpr int PAIR_HEAVY = 2048;
struct PairPass {
    const int32_t* __restrict__ rank_pair;
    const int32_t* __restrict__ pair_start;
    const int32_t* __restrict__ pair_di;
    const int32_t* __restrict__ csc_slot;
    const float* __restrict__ csc_r;
    const double2* __restrict__ ud_slot;   // [slot] = (s_v, n_v): ONE 16-byte gather per entry (the pass runs at the L2's request rate)
    int32_t lo, hi;
    // x = r / s_v per CSC entry (the walk's weights, fy_cooc.hpp) written on the way -- the gather of s_v is this pass's anyway (round 4:
    // a kernel of its own, k_csc_x, with the same 25 M gathers); x / s_v only for the packed walk (null otherwise)
    float* __restrict__ csc_x;
    float* __restrict__ csc_x_over_s;
    double* __restrict__ partial;      // by dense item (several clusters may add to one item)
    double* __restrict__ b_rank;       // by rank position
    long long* __restrict__ walk_rank; // by rank position: sum of the raters' degrees
    int32_t* __restrict__ cnt_rank;    // by rank position: raters
    float* __restrict__ fx_rank;       // by rank position, 3 floats: sum and maximum of r / s_v^2 over the column, maximum rating
};                                     // (bounds of the fixed-point scale of the row kernel, fx_exponent)
struct PairAcc {   // (no member initialisers: instances live in __shared__ memory too)
    double ps, b, ws;
    long long w;
    float wm, rm;
};
__device__ __forceinline__ PairAcc fy_pair_zero() { return PairAcc{0.0, 0.0, 0.0, 0, 0.0f, 0.0f}; }
__device__ __forceinline__ void fy_pair_entry(const PairPass& A, int32_t q, PairAcc& a) {
    const int32_t slot = A.csc_slot[q];
    const double r = (double)A.csc_r[q];
    if (slot >= A.lo && slot < A.hi) a.ps += r;
    const double2 ud = A.ud_slot[slot];
    const double inv = 1.0 / ud.x;                    // ONE fp64 division per entry for the column sums (round 4: r / s and r / (s * s) were two;
    const double x = r * inv;                         // the heavy-column kernel ran with 40 spilled VGPRs around them)
    if (A.csc_x || A.csc_x_over_s) {                  // (kernel-uniform) the stored weights keep their own roundings: r / s, then / s
        const double xd = r / ud.x;
        if (A.csc_x) A.csc_x[q] = (float)xd;
        if (A.csc_x_over_s) A.csc_x_over_s[q] = (float)(xd / ud.x);
    }
    a.b += x;
    a.w += (long long)ud.y;
    const double wt = x * inv;
    a.ws += wt;
    a.wm = fmaxf(a.wm, (float)wt);
    a.rm = fmaxf(a.rm, (float)r);
}
template <int G = 64>
__device__ __forceinline__ void fy_pair_reduce(PairAcc& a) {      // over groups of G consecutive lanes
    for (int o = G / 2; o > 0; o >>= 1) {
        a.ps += __shfl_down(a.ps, o, 64);
        a.b += __shfl_down(a.b, o, 64);
        a.ws += __shfl_down(a.ws, o, 64);
        a.w += __shfl_down(a.w, o, 64);
        a.wm = fmaxf(a.wm, __shfl_down(a.wm, o, 64));
        a.rm = fmaxf(a.rm, __shfl_down(a.rm, o, 64));
    }
}
__device__ __forceinline__ void fy_pair_store(const PairPass& A, int32_t pos, int32_t pr, int32_t n, const PairAcc& a) {
    if (a.ps != 0.0) atomicAdd(&A.partial[A.pair_di[pr]], a.ps);
    A.b_rank[pos] = a.b;
    A.walk_rank[pos] = a.w;
    A.cnt_rank[pos] = n;
    A.fx_rank[3 * (int64_t)pos + 0] = (float)(a.ws * 1.000001);      // rounded up: these are upper bounds
    A.fx_rank[3 * (int64_t)pos + 1] = a.wm * 1.000001f;
    A.fx_rank[3 * (int64_t)pos + 2] = a.rm;
}
// G lanes per column: 64 for the long columns of a big cluster; 16 where the average column is short (50 clusters of ML-25M shape:
// 2.5 M columns of 10 raters -- a wave per column spent 1.7 ms per job there, four columns per wave 0.5)
// heavy columns are cut into chunks of PAIR_CHUNK entries, one workgroup per chunk (k_pair_chunks), summed per column in chunk order
// (k_pair_finish).  Up to round 4 a workgroup of 1024 walked a whole heavy column: the 10^5-rater columns of ML-25M shape made
// that 0.34 ms of latency-bound tail.
constexpr int PAIR_CHUNK = 4096;
struct PairHeavy {
    int32_t* __restrict__ col;        // [k]: rank position of heavy column k
    int32_t* __restrict__ base;       // [k]: its first chunk
    int32_t* __restrict__ chunk_col;  // [chunk]: k
    int32_t* __restrict__ counters;   // [0] heavy columns, [1] chunks
    PairAcc* __restrict__ acc;        // [chunk]
};
template <int G>
__global__ void k_pair_pass(int32_t nP, PairPass A, PairHeavy H) {
    const int lane = threadIdx.x & (G - 1), gpb = blockDim.x / G;
    const int32_t stride = gridDim.x * gpb;
    // (every group of a wave runs the same number of rounds: the shuffles of the reduction need all lanes)
    for (int32_t pos0 = blockIdx.x * gpb; pos0 < nP; pos0 += stride) {
        const int32_t pos = pos0 + (int32_t)(threadIdx.x / G);
        const bool live = pos < nP;
        const int32_t pr = live ? A.rank_pair[pos] : 0;
        const int32_t q0 = live ? A.pair_start[pr] : 0, q1 = live ? A.pair_start[pr + 1] : 0;
        const bool is_heavy = q1 - q0 > PAIR_HEAVY;
        if (is_heavy && lane == 0) {
            const int32_t nchk = (q1 - q0 + PAIR_CHUNK - 1) / PAIR_CHUNK;
            const int32_t k = atomicAdd(&H.counters[0], 1), c0 = atomicAdd(&H.counters[1], nchk);
            H.col[k] = pos;
            H.base[k] = c0;
            for (int32_t i = 0; i < nchk; i++) H.chunk_col[c0 + i] = k;
        }
        PairAcc a = fy_pair_zero();
        if (!is_heavy)
            for (int32_t q = q0 + lane; q < q1; q += G) fy_pair_entry(A, q, a);
        fy_pair_reduce<G>(a);
        if (live && !is_heavy && lane == 0) fy_pair_store(A, pos, pr, q1 - q0, a);
    }
}
__device__ __forceinline__ void fy_pair_add(PairAcc& t, const PairAcc& x) {
    t.ps += x.ps; t.b += x.b; t.ws += x.ws; t.w += x.w;
    t.wm = fmaxf(t.wm, x.wm); t.rm = fmaxf(t.rm, x.rm);
}
__global__ __launch_bounds__(256) void k_pair_chunks(PairPass A, PairHeavy H) {
    __shared__ PairAcc sh_a[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = H.counters[1];
    for (int c = blockIdx.x; c < n; c += gridDim.x) {
        const int32_t k = H.chunk_col[c];
        const int32_t pr = A.rank_pair[H.col[k]];
        const int32_t q0 = A.pair_start[pr] + (c - H.base[k]) * PAIR_CHUNK, q1 = min(A.pair_start[pr + 1], q0 + PAIR_CHUNK);
        PairAcc a = fy_pair_zero();
#pragma unroll 4
        for (int32_t q = q0 + (int32_t)threadIdx.x; q < q1; q += 256) fy_pair_entry(A, q, a);
        fy_pair_reduce(a);
        if (lane == 0) sh_a[wave] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            PairAcc t = sh_a[0];
            for (int x = 1; x < 4; x++) fy_pair_add(t, sh_a[x]);
            H.acc[c] = t;
        }
        __syncthreads();
    }
}
__global__ void k_pair_finish(PairPass A, PairHeavy H) {
    const int n = H.counters[0];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int32_t pos = H.col[k], pr = A.rank_pair[pos];
        const int32_t len = A.pair_start[pr + 1] - A.pair_start[pr], nchk = (len + PAIR_CHUNK - 1) / PAIR_CHUNK;
        PairAcc t = H.acc[H.base[k]];
        for (int32_t i = 1; i < nchk; i++) fy_pair_add(t, H.acc[H.base[k] + i]);      // chunk order: the same sum in every run
        fy_pair_store(A, pos, pr, len, t);
    }
}
// per-slot copies of the user sums and degrees: one gather per CSC entry instead of two dependent ones
__global__ void k_slot_user_arrays(int32_t nU, const int32_t* __restrict__ slot2du, const double* __restrict__ usum,
                                   const int32_t* __restrict__ udeg, double* __restrict__ usum_slot, int32_t* __restrict__ deg_slot,
                                   double2* __restrict__ ud_slot) {
    for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nU; s += gridDim.x * blockDim.x) {
        const int32_t du = slot2du[s];
        usum_slot[s] = usum[du];
        deg_slot[s] = udeg[du];
        ud_slot[s] = make_double2(usum[du], (double)udeg[du]);
    }
}

// per cluster: maxima over its items of (sum of r / s^2, largest r / s^2, largest rating) -> bounds of a Gram entry and of one contribution
__global__ void k_cluster_fx_bounds(const int32_t* __restrict__ pcstart, const float* __restrict__ fx_rank, float* __restrict__ out) {
    // blockIdx.x = cluster, blockIdx.y = share of its items; `out` is zeroed by the caller and the values are >= 0, so the maximum of
    // their bit patterns is their maximum (one cluster of 59 047 items in ONE workgroup was 67 us of every prepare)
    const int c = blockIdx.x;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    for (int32_t pos = pcstart[c] + blockIdx.y * blockDim.x + threadIdx.x; pos < pcstart[c + 1]; pos += gridDim.y * blockDim.x) {
        m0 = fmaxf(m0, fx_rank[3 * (int64_t)pos]);
        m1 = fmaxf(m1, fx_rank[3 * (int64_t)pos + 1]);
        m2 = fmaxf(m2, fx_rank[3 * (int64_t)pos + 2]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        m0 = fmaxf(m0, __shfl_down(m0, o, 64));
        m1 = fmaxf(m1, __shfl_down(m1, o, 64));
        m2 = fmaxf(m2, __shfl_down(m2, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (m0 > 0.f) atomicMax(reinterpret_cast<int*>(out + 3 * c), __float_as_int(m0));
        if (m1 > 0.f) atomicMax(reinterpret_cast<int*>(out + 3 * c + 1), __float_as_int(m1));
        if (m2 > 0.f) atomicMax(reinterpret_cast<int*>(out + 3 * c + 2), __float_as_int(m2));
    }
}

// quirk Q1: the Hadoop counter adds (long) s_u * 100 per user and is divided by 100 afterwards
// (DoubleSumAndCountReducer.java:41, RM2Job.java:95): the total is sum_u floor(s_u).
__global__ void k_partial_total(int32_t lo, int32_t hi, const int32_t* __restrict__ slot2du, const double* __restrict__ usum,
                                unsigned long long* __restrict__ counter) {
    unsigned long long local = 0;
    for (int32_t s = lo + blockIdx.x * blockDim.x + threadIdx.x; s < hi; s += gridDim.x * blockDim.x)
        local += (unsigned long long)((long long)usum[slot2du[s]] * 100LL);
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    __shared__ unsigned long long sh[16];      // one atomic per workgroup (2 540 waves on one address were most of this kernel's 33 us)
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) local += sh[w];
        if (local) atomicAdd(counter, local);
    }
}
__global__ void k_store_total(const unsigned long long* __restrict__ counter, double* __restrict__ dst) {
    *dst = (double)*counter;   // exact below 2^53; the division by OFFSET = 100 happens after the exchange
}

// fixed rank order => bit-reproducible regardless of how the all-gather was scheduled
__global__ void k_sum_gathered(int64_t len, int32_t world, const double* __restrict__ gathered, double* __restrict__ out) {
    for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d < len; d += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int r = 0; r < world; r++) s += gathered[(int64_t)r * len + d];
        out[d] = s;
    }
}

// Sharded prep (whole clusters per rank, only the owned clusters' ratings prepared): a rank's dense item / user numbering is its own,
// so the exchange buffer is laid out by RAW id -- [max_item + 1 item sums][counter][max_user + 1 user sums][failure flag] -- and the
// job's own dense statistics are read back out of the sum over ranks.
__global__ void k_stats_to_raw(int32_t nI, const int32_t* __restrict__ iid, const double* __restrict__ partial, int64_t n_raw_items, int32_t nU,
                               const int32_t* __restrict__ uid, const double* __restrict__ usum, double* __restrict__ raw) {
    const int32_t n = max(nI + 1, nU);
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        if (t < nI) raw[iid[t]] = partial[t];
        if (t == nI) raw[n_raw_items] = partial[nI];
        if (t < nU) raw[n_raw_items + 1 + uid[t]] = usum[t];
    }
}
__global__ void k_stats_from_raw(int32_t nI, const int32_t* __restrict__ iid, const double* __restrict__ raw, int64_t n_raw_items,
                                 double* __restrict__ stats) {
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t <= nI; t += gridDim.x * blockDim.x) stats[t] = t < nI ? raw[iid[t]] : raw[n_raw_items];
}
__global__ void k_rank_flags(int32_t world, int64_t len, const double* __restrict__ gathered, double* __restrict__ out) {
    if ((int)threadIdx.x < world) out[threadIdx.x] = gathered[(int64_t)threadIdx.x * len + len - 1];
}
__global__ void k_flag_positive(int64_t n, const double* __restrict__ v, uint32_t* __restrict__ flag) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) flag[t] = v[t] > 0.0 ? 1u : 0u;
}
// (id, value * scale) of the positive entries, ascending id
__global__ void k_compact_positive(int64_t n, const double* __restrict__ v, const uint32_t* __restrict__ pos, const double* __restrict__ divide_by,
                                   int32_t* __restrict__ id, double* __restrict__ val) {
    const double d = divide_by ? *divide_by : 1.0;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        if (v[t] > 0.0) {
            id[pos[t] - 1] = (int32_t)t;
            val[pos[t] - 1] = divide_by ? v[t] / d : v[t];
        }
}

// p(i|C) = itemsum / totalSum (DoubleSumAndDividerReducer.java:35-44); stats[nI] holds the counter (x100)
__global__ void k_item_coll(int32_t nI, const double* __restrict__ stats, double* __restrict__ icoll, double* __restrict__ total_out) {
    const double total = stats[nI] / 100.0;
    for (int32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < nI; d += gridDim.x * blockDim.x) icoll[d] = stats[d] / total;
    if (blockIdx.x == 0 && threadIdx.x == 0) *total_out = total;
}

// p(i|C) of every (cluster, item) in rank order, a = l * p, b rounded once to fp32
__global__ void k_pair_p(int32_t nP, const int32_t* __restrict__ rank_pair, const int32_t* __restrict__ pair_di,
                         const double* __restrict__ icoll, double lambda, const double* __restrict__ b_rank,
                         double* __restrict__ p_rank, float* __restrict__ a_rank, float* __restrict__ b_rank32, double2* __restrict__ pb_rank) {
    for (int32_t pos = blockIdx.x * blockDim.x + threadIdx.x; pos < nP; pos += gridDim.x * blockDim.x) {
        const double p = icoll[pair_di[rank_pair[pos]]];
        p_rank[pos] = p;
        pb_rank[pos] = make_double2(p, b_rank[pos]);      // (p, b) side by side: k_csr_values gathers both with one 16-byte load per rating
        a_rank[pos] = (float)(lambda * p);
        b_rank32[pos] = (float)b_rank[pos];
    }
}

// x = r / s_v per CSC entry; for the packed row kernel also x / s_v (the CSR side then carries the raw rating)
__global__ void k_csc_x(int64_t nnz, const int32_t* __restrict__ csc_slot, const float* __restrict__ csc_r,
                        const double* __restrict__ usum_slot, float* __restrict__ csc_x, float* __restrict__ csc_x_over_s) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x) {
        const double s = usum_slot[csc_slot[q]];
        const double x = (double)csc_r[q] / s;
        csc_x[q] = (float)x;
        if (csc_x_over_s) csc_x_over_s[q] = (float)(x / s);
    }
}

// row (popularity rank inside its cluster) of every CSC entry: the symmetric walk cuts a rater's slice behind it
__global__ void k_csc_rank(int64_t nnz, const int32_t* __restrict__ csc_pair, const int32_t* __restrict__ pair_rank, int32_t* __restrict__ out) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x) out[q] = pair_rank[csc_pair[q]];
}

// packed CSR of one cluster for the row kernel (fy_cooc.hpp): column index relative to its chunk | fp16 raw rating
__device__ __forceinline__ void pack_csr_body(int32_t f0, int32_t f1, int32_t CH, const int32_t* __restrict__ csr_idx, const float* __restrict__ csr_r,
                                              uint32_t* __restrict__ pk) {
    for (int64_t f = f0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < f1; f += (int64_t)gridDim.x * blockDim.x)
        pk[f] = (uint32_t)(csr_idx[f] % CH) | ((uint32_t)__half_as_ushort(__float2half(csr_r[f])) << 16);
}
__global__ void k_pack_csr(int32_t f0, int32_t f1, int32_t CH, const int32_t* __restrict__ csr_idx, const float* __restrict__ csr_r,
                           uint32_t* __restrict__ pk) {
    pack_csr_body(f0, f1, CH, csr_idx, csr_r, pk);
}

// Column-panel mode, tail rows (rows >= p_eff, few raters each): their co-ratings with the columns behind p_eff are needed only
// as upper bounds of the 64-column block maxima.  For a row i and a block B
//     max_{j in B} sum_v x_vi x_vj  <=  sum_v x_vi max_{j in B} x_vj ,
// and the right-hand side is the row kernel's own sum over a CSR compressed to ONE entry per (rater, block): the block index
// relative to p_eff | the largest raw rating of the rater in the block, in the packed format of k_pack_csr.  With 2-17 raters
// per tail row the sum of maxima is within ~14 % of the exact maximum (measured at ML-25M shape in 50 clusters: 1088
// surviving blocks instead of 1080), a row needs ONE item with (I_c - p_eff) / 64 accumulators instead of one item per
// 8192-column chunk, and no matrix element behind the panel is ever formed.
// The user's compressed entries are written in place of the first entries of its CSR range (y_pk is as long as the CSR),
// co2[2 k] / co2[2 k + 1] = their range: the "chunk offsets" of a one-chunk segment table.
__device__ __forceinline__ void tail_blocks_body(int32_t slot_base, int32_t n_slots, int32_t p_eff, const int32_t* __restrict__ rowptr,
                                                const int32_t* __restrict__ csr_idx, const float* __restrict__ csr_r, uint32_t* __restrict__ y_pk,
                                                int32_t* __restrict__ co2) {
    // One workgroup per user: the entries behind p_eff are a suffix of the row (found by a binary search), its 64-entry pieces
    // are dealt to the four waves; a piece looks one entry back for the block of its predecessor and takes its output slots
    // from an LDS counter (the order of a user's compressed entries does not matter: the sums they enter are integer sums).
    __shared__ int sh_count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int32_t k = blockIdx.x; k < n_slots; k += gridDim.x) {      // block-uniform
        const int32_t base = rowptr[slot_base + k], end = rowptr[slot_base + k + 1];
        int32_t lo = base, hi = end;                                  // first entry with column >= p_eff
        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (csr_idx[mid] < p_eff) lo = mid + 1; else hi = mid; }
        const int32_t first = lo;
        if (threadIdx.x == 0) sh_count = 0;
        __syncthreads();
        for (int32_t f0 = first + wave * 64; f0 < end; f0 += 4 * 64) {   // wave-uniform
            const int32_t f = f0 + lane;
            const int blk = f < end ? (csr_idx[f] - p_eff) >> 6 : -1;
            int prev = __shfl_up(blk, 1, 64);
            if (lane == 0) prev = f0 > first ? (csr_idx[f0 - 1] - p_eff) >> 6 : -1;
            const bool start = blk >= 0 && blk != prev;
            const unsigned long long bal = __ballot(start);
            int at = 0;
            if (lane == 0 && bal) at = atomicAdd(&sh_count, (int)__popcll(bal));
            at = __shfl(at, 0, 64);
            if (start) {
                float m = csr_r[f];
                for (int32_t g = f + 1; g < end && ((csr_idx[g] - p_eff) >> 6) == blk; g++) m = fmaxf(m, csr_r[g]);
                y_pk[base + at + __popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)blk | ((uint32_t)__half_as_ushort(__float2half(m)) << 16);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { co2[2 * k] = base; co2[2 * k + 1] = base + sh_count; }
        __syncthreads();     // sh_count is read before the next user resets it
    }
}
__global__ __launch_bounds__(256) void k_tail_blocks(int32_t slot_base, int32_t n_slots, int32_t p_eff, const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ csr_idx, const float* __restrict__ csr_r, uint32_t* __restrict__ y_pk,
                                                     int32_t* __restrict__ co2) {
    tail_blocks_body(slot_base, n_slots, p_eff, rowptr, csr_idx, csr_r, y_pk, co2);
}

// one wave per user row: x = r / s_u and e = (1-l)(b_j - x) + l (U_c - 1) p_j  (all fp64, rounded once)
__global__ void k_csr_values(int32_t nU, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr_idx,
                             const float* __restrict__ csr_r, const int32_t* __restrict__ slot2du,
                             const int32_t* __restrict__ ucluster, const double* __restrict__ usum,
                             const int32_t* __restrict__ csize, const int32_t* __restrict__ pcstart,
                             const double2* __restrict__ pb_rank, double lambda,
                             const float* __restrict__ gscale /* [cluster]: 2^-c of the packed matrix format, 1 for fp32 rows */,
                             float* __restrict__ csr_x, float* __restrict__ csr_e, float* __restrict__ csr_q) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int32_t s = blockIdx.x * wpb + (threadIdx.x >> 6); s < nU; s += gridDim.x * wpb) {
        const int32_t du = slot2du[s];
        const int32_t c = ucluster[du];
        const double sum = usum[du];
        const double Uc1 = (double)(csize[c] - 1);
        const double gs = (double)gscale[c];
        const int32_t pb = pcstart[c];
        for (int32_t f = rowptr[s] + lane; f < rowptr[s + 1]; f += 64) {
            const int32_t j = csr_idx[f];
            const double x = (double)csr_r[f] / sum;
            const double2 pbj = pb_rank[pb + j];       // (p(j|C), b_j)
            double e = (1.0 - lambda) * (pbj.y - x) + lambda * Uc1 * pbj.x;
            if (!(e > 0.0)) e = 0.0;
            csr_x[f] = (float)x;
            csr_e[f] = (float)(e * gs);
            csr_q[f] = (float)(lambda * (1.0 - lambda) * pbj.x * gs);     // q_j: the rank-one part of a term is q_j b_i
        }
    }
}

// per slot of this rank: pvpi, whether the user gets a list, how many rows it emits
__global__ void k_user_meta(int32_t lo, int32_t hi, const int32_t* __restrict__ slot2du, const int32_t* __restrict__ uid,
                            const int32_t* __restrict__ ucluster, const int32_t* __restrict__ udeg,
                            const int32_t* __restrict__ csize, const int32_t* __restrict__ pcstart, int32_t number_of_items,
                            int32_t top_n, int32_t filter_users, const int32_t* __restrict__ cshift /* [cluster]: c of the packed format */,
                            double* __restrict__ pvpi, int32_t* __restrict__ n_out,
                            unsigned long long* __restrict__ counters /* [0] log terms, [1] users scored */) {
    unsigned long long terms = 0, scored = 0;
    for (int32_t s = lo + blockIdx.x * blockDim.x + threadIdx.x; s < hi; s += gridDim.x * blockDim.x) {
        const int32_t du = slot2du[s];
        const int32_t c = ucluster[du];
        const int32_t n = udeg[du];
        const int32_t Ic = pcstart[c + 1] - pcstart[c];
        const int32_t unrated = Ic - n;
        // AbstractRM2Reducer.java:327-329
        // (+ n c ln 2: the n log terms of a user of a cluster with a scaled matrix are each short by c bits, see FY_P24_SHIFT)
        pvpi[s - lo] = (double)(n - 1) * log((double)number_of_items) - (double)n * log((double)csize[c]) +
                       (double)n * (double)cshift[c] * 0.69314718055994530942;
        // :210-213 (no unrated item -> skipped with a warning), :221-223 (filterUsers)
        const bool skip = unrated <= 0 || uid[du] < filter_users;
        int32_t k = skip ? 0 : min(top_n, unrated);
        if (k < 0) k = 0;
        n_out[s - lo] = k;
        if (!skip) { terms += (unsigned long long)n * (unsigned long long)unrated; scored++; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        terms += __shfl_down(terms, o, 64);
        scored += __shfl_down(scored, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (terms) atomicAdd(&counters[0], terms);
        if (scored) atomicAdd(&counters[1], scored);
    }
}

// ================================================================ chunk offsets for the row kernel
__device__ __forceinline__ void chunk_offsets_body(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr_idx, int32_t slot_base,
                                                   int32_t n_slots, int32_t CH, int32_t nch, int32_t* __restrict__ chunk_off) {
    const int64_t total = (int64_t)n_slots * (nch + 1);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t v = (int32_t)(t / (nch + 1)), ch = (int32_t)(t % (nch + 1));
        const int32_t a = rowptr[slot_base + v], b = rowptr[slot_base + v + 1];
        int32_t lo = a, hi = b;
        const int32_t key = ch * CH;   // first entry with idx >= key
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (csr_idx[mid] < key) lo = mid + 1; else hi = mid;
        }
        chunk_off[t] = (ch == nch) ? b : lo;
    }
}
__global__ void k_chunk_offsets(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr_idx, int32_t slot_base,
                                int32_t n_slots, int32_t CH, int32_t nch, int32_t* __restrict__ chunk_off) {
    chunk_offsets_body(rowptr, csr_idx, slot_base, n_slots, CH, nch, chunk_off);
}

void build_chunk_offsets(Context* ctx, const int32_t* rowptr, const int32_t* csr_idx, int32_t slot_base, int32_t n_slots,
                         int32_t CH, int32_t nch, int32_t* chunk_off, hipStream_t st) {
    const int64_t total = (int64_t)n_slots * (nch + 1);
    if (total == 0) return;
    k_chunk_offsets<<<grid_for(total), 256, 0, st ? st : ctx->stream>>>(rowptr, csr_idx, slot_base, n_slots, CH, nch, chunk_off);
    FY_KERNEL_CHECK();
}

// ---- segment table (fy_cooc.hpp): counts, exclusive prefix, fill
// HALF mode (symmetric walk): the co-rating Gram is symmetric, so row i accumulates only the columns j > i and a mirror pass
// fills the lower triangle (k_mirror_*).  For the CSC entry (rater v, row i) the chunks in front of i's own chunk have no
// segments, and in i's chunk the rater's slice starts behind the position of i in the rater's CSR row (`Half::start`,
// found once by a binary search in k_seg_counts and re-used by k_seg_fill).
#ifndef FY_SAMPLE_SHIFT
#define FY_SAMPLE_SHIFT 6     // the symmetric cut searches every 64th CSR entry (measured, ML-25M job, tables / row kernel ms: exact cut 4.57 / 7.84; stride 64: 2.96 / 8.48 = 25.6 ms per job; stride 16: 3.23 / 8.21 = 25.7; Netflix shape 77.1 / 79.1 ms per job)
#endif
struct Half {
    const int32_t* __restrict__ row_of_entry;   // [q0 + q]: the row (rank inside the cluster) of CSC entry q; nullptr = full walk
    const int32_t* __restrict__ csr_idx;
    int32_t CH;
    int32_t* __restrict__ start;                // [q]: where the slice of CSC entry q starts in the row's own chunk: the 64-entry-aligned CSR
                                                // position in front of (v, i) -- the row kernel masks the entries at or before column i
    const int32_t* __restrict__ samp;           // [k] = csr_idx[64 k]: every 64th CSR entry (k_csr_samples)
    // tail-row launches (column-panel mode): entries of the rows in front of only_from get no segments
    const int32_t* __restrict__ only_rows;      // [q0 + q]: row of the entry; nullptr = all rows
    int32_t only_from;
    // column-panel mode: only the rows in front of sym_rows are cut (0 = all rows, plain half walk; < 0 = none: row_of_entry is set
    // for the chunk limit alone), and the rows from lim_from on have segments only in their first lim_chunks chunks (0 = off) --
    // rows and chunks the item list never names (tail rows behind the panel's chunks; symmetric panel mode: every row)
    int32_t sym_rows, lim_from, lim_chunks;
};
__device__ __forceinline__ void seg_counts_body(const int32_t* __restrict__ csc_slot, const int32_t* __restrict__ chunk_off, int32_t slot_base,
                                                int32_t q0, int32_t nq, int32_t nch, int32_t* __restrict__ cnt, const Half& H) {
    // one thread per CSC entry: the rater's nch + 1 chunk offsets are one contiguous gather -- ONE 16-byte load where a row of the
    // table is four offsets (three chunks: ML-25M shape).  Round 4: the kernel ran at 72 % of the L2's request rate (6.5 requests per
    // entry, TCP_TCC_READ_REQ), most of them the same 16 bytes fetched offset by offset from 64 different rows per wave instruction.
    const bool row4 = nch == 3 && (reinterpret_cast<uintptr_t>(chunk_off) & 15) == 0;      // (kernel-uniform)
    const bool row2 = nch == 1 && (reinterpret_cast<uintptr_t>(chunk_off) & 7) == 0;       // one chunk (Netflix shape): two offsets, 8 bytes
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q <= nq; q += (int64_t)gridDim.x * blockDim.x) {
        const bool live = q < nq && !(H.only_rows && H.only_rows[q0 + q] < H.only_from);
        const int32_t* co = live ? chunk_off + (int64_t)(csc_slot[q0 + q] - slot_base) * (nch + 1) : nullptr;
        int4 c4 = make_int4(0, 0, 0, 0);
        if (live && row4) c4 = *reinterpret_cast<const int4*>(co);
        if (live && row2) { const int2 v = *reinterpret_cast<const int2*>(co); c4.x = v.x; c4.y = v.y; }
        auto CO = [&](int k) -> int32_t { return (row4 || row2) ? (k == 0 ? c4.x : k == 1 ? c4.y : k == 2 ? c4.z : c4.w) : co[k]; };
        int32_t prev = live ? CO(0) : 0;
        int32_t own = -1, behind = 0, climit = nch;
        const int32_t r_e = (live && H.row_of_entry) ? H.row_of_entry[q0 + q] : 0;
        if (live && H.row_of_entry && H.lim_chunks > 0 && r_e >= H.lim_from) climit = H.lim_chunks;
        if (live && H.row_of_entry && (H.sym_rows == 0 || r_e < H.sym_rows)) {
            const int32_t r = r_e;
            own = r / H.CH;
            // Where does the slice behind (v, i) start?  Exactly: behind the position of i in the rater's row -- a binary search over the
            // row slice in the 100-400 MB csr_idx per CSC entry (1.6 ms of the ML-25M job, 10.6 ms at Netflix shape, round 2).  Now: the
            // last aligned CSR position inside the slice whose entry is <= r, found in the SAMPLES (every 2^FY_SAMPLE_SHIFT-th entry: a few MB,
            // cache-resident, a handful of probes); the few entries between it and i are masked by the row kernel (column <= row).
            constexpr int SS = FY_SAMPLE_SHIFT;
            const int32_t co_own = CO(own), co_next = CO(own + 1);
            int32_t klo = (co_own + (1 << SS) - 1) >> SS, khi = (co_next + (1 << SS) - 1) >> SS;      // samples inside the slice: [klo, khi)
            while (klo < khi) {                                                    // first k with samp[k] > r
                const int32_t mid = (klo + khi) >> 1;
                if (H.samp[mid] <= r) klo = mid + 1; else khi = mid;
            }
            const int32_t lo = max(co_own, (klo - 1) << SS);                      // (no sample <= r inside the slice: its beginning)
            behind = lo;
            H.start[q] = lo;
        }
        for (int32_t ch = 0; ch < nch; ch++) {
            int32_t n = 0;
            if (live) {
                const int32_t next = CO(ch + 1);
                const int32_t first = ch == own ? behind : prev;
                n = (ch < own || ch >= climit) ? 0 : (next - first + 63) >> 6;
                prev = next;
            }
            cnt[(int64_t)ch * (nq + 1) + q] = n;
        }
    }
}
__global__ void k_seg_counts(const int32_t* __restrict__ csc_slot, const int32_t* __restrict__ chunk_off, int32_t slot_base,
                             int32_t q0, int32_t nq, int32_t nch, int32_t* __restrict__ cnt, Half H) {
    seg_counts_body(csc_slot, chunk_off, slot_base, q0, nq, nch, cnt, H);
}
// The tables of ALL clusters of a job in one pass (many clusters, the reference's regime): blockIdx.y = table.  A table's counts /
// prefixes are a range of one array (cnt_off), so ONE scan numbers the segments of all tables and one fill writes them -- built
// cluster by cluster the same 25 M CSC entries cost ~15 small launches and a host round trip per cluster: 20 ms at 50 clusters
// against 4.6 ms for the one-cluster job.
struct SegDesc {
    int32_t slot_base, q0, nq, nch, CH, half, only_from, co_stride;
    int64_t co_off, cnt_off;      // first entry of the table's chunk offsets / of its counts and prefixes
    int32_t use_only_rows, sym_rows;      // sym_rows, lim_from, lim_chunks: Half
    int32_t lim_from, lim_chunks;
};
__global__ void k_seg_counts_multi(const SegDesc* __restrict__ D, const int32_t* __restrict__ csc_slot, const int32_t* __restrict__ co_all,
                                   int32_t* __restrict__ cnt_all, const int32_t* __restrict__ row_of_entry, const int32_t* __restrict__ csr_idx,
                                   int32_t* __restrict__ start_all, const int32_t* __restrict__ samples) {
    const SegDesc d = D[blockIdx.y];
    const Half H{(d.half || d.lim_chunks > 0) ? row_of_entry : nullptr, csr_idx, d.CH, start_all + d.q0, samples, d.use_only_rows ? row_of_entry : nullptr, d.only_from,
                 d.half ? d.sym_rows : -1, d.lim_from, d.lim_chunks};
    seg_counts_body(csc_slot, co_all + d.co_off, d.slot_base, d.q0, d.nq, d.nch, cnt_all + d.cnt_off, H);
}

// One wave fills the segments of 64 consecutive CSC entries of one chunk cooperatively: the group's segments are
// contiguous in the table (exclusive prefix `ptr`), lane l writes segment base + l, base + l + 64, ... after finding its
// owner among the 64 entries with a binary search over shuffled prefix values -- every store of the 3 GB table is
// coalesced (one thread per entry writing its own run of segments reached 1.1 TB/s).
__device__ __forceinline__ void seg_fill_body(const int32_t* __restrict__ csc_slot, const float* __restrict__ csc_w, const int32_t* __restrict__ chunk_off,
                                              int32_t slot_base, int32_t q0, int32_t nq, int32_t nch, const int32_t* __restrict__ ptr,
                                              int2* __restrict__ seg, float* __restrict__ seg_w, const Half& H) {
    // A wave takes the group's entries through up to CB chunks at once: what is per ENTRY (rater slot, row, weight) is loaded once, and the
    // loads that are per (entry, chunk) -- chunk offsets, segment prefixes -- are all in flight before the first segment is written.
    // (Round 4: one (group, chunk) per wave iteration meant one dependent chain slot -> chunk offsets -> prefix per ~100 segments: the fill
    // ran at 1.4 TB/s of stores, bound by that latency.)
    constexpr int CB = 4;
    const bool row4 = nch == 3 && (reinterpret_cast<uintptr_t>(chunk_off) & 15) == 0;      // a table row = four offsets = one 16-byte load
    const bool row2 = nch == 1 && (reinterpret_cast<uintptr_t>(chunk_off) & 7) == 0;       // two offsets = one 8-byte load
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int64_t n_groups = ((int64_t)nq + 63) >> 6;
    const int32_t n_cb = (nch + CB - 1) / CB;
    const int64_t total_work = n_groups * n_cb;
    for (int64_t gw = blockIdx.x * (int64_t)wpb + (threadIdx.x >> 6); gw < total_work; gw += (int64_t)gridDim.x * wpb) {
        const int32_t ch0 = (int32_t)(gw / n_groups) * CB;
        const int64_t g = gw - (int64_t)(ch0 / CB) * n_groups;
        const int64_t q = g * 64 + lane;
        const int64_t q_end = min((int64_t)nq, g * 64 + 64);
        const bool live = q < nq;
        int32_t cov[CB + 1], startv[CB], endv[CB];
        int32_t r = 0, own = -1, own_start = 0;
        bool dead = false, limited = false;
        float w = 0.0f;
#pragma unroll
        for (int c = 0; c < CB; c++) {      // (wave-uniform tests)
            startv[c] = 0;
            endv[c] = ch0 + c < nch ? ptr[(int64_t)(ch0 + c) * (nq + 1) + q_end] : 0;
            if (live && ch0 + c < nch) startv[c] = ptr[(int64_t)(ch0 + c) * (nq + 1) + q];
        }
#pragma unroll
        for (int c = 0; c <= CB; c++) cov[c] = 0;
        if (live) {
            const int32_t* co = chunk_off + (int64_t)(csc_slot[q0 + q] - slot_base) * (nch + 1) + ch0;
            if (row4) {
                const int4 v = *reinterpret_cast<const int4*>(co);      // (ch0 = 0: the whole row)
                cov[0] = v.x; cov[1] = v.y; cov[2] = v.z; cov[3] = v.w;
            } else if (row2) {
                const int2 v = *reinterpret_cast<const int2*>(co);
                cov[0] = v.x; cov[1] = v.y;
            } else {
#pragma unroll
                for (int c = 0; c <= CB; c++)
                    if (ch0 + c <= nch) cov[c] = co[c];
            }
            if (H.row_of_entry) {
                r = H.row_of_entry[q0 + q];
                if (H.sym_rows == 0 || r < H.sym_rows) {
                    own = r / H.CH;
                    if (own >= ch0 && own < ch0 + CB) own_start = H.start[q];
                }
                limited = H.lim_chunks > 0 && r >= H.lim_from;
            }
            if (H.only_rows && H.only_rows[q0 + q] < H.only_from) dead = true;
            w = csc_w[q0 + q];
        }
#pragma unroll
        for (int c = 0; c < CB; c++) {
            const int32_t ch = ch0 + c;
            if (ch >= nch) break;                             // wave-uniform
            int32_t f0 = cov[c], len = cov[c + 1] - cov[c], start = startv[c];
            if (live) {
                if (own >= 0) {
                    if (ch < own) len = 0;
                    else if (ch == own) { f0 = own_start; len = cov[c + 1] - f0; }
                }
                if (limited && ch >= H.lim_chunks) len = 0;
                if (dead) len = 0;
            } else len = 0;
            const int32_t base = __shfl(start, 0, 64);
            const int32_t total = endv[c] - base;                 // segments of the whole group
            if (!live) start = base + total;                      // inactive lanes sit behind the last segment
            const int32_t rel = start - base;
            for (int32_t kk = 0; kk < total; kk += 64) {         // wave-uniform trip count: the shuffles need every lane alive
                const int32_t k = kk + lane;
                int lo = 0, hi = 63;                              // largest lane j with rel_j <= k (rel is non-decreasing)
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const int mid = (lo + hi + 1) >> 1;
                    const int32_t rm = __shfl(rel, mid, 64);
                    if (rm <= k) lo = mid; else hi = mid - 1;
                }
                // entries with zero segments share their rel with the next entry: the owner is the LAST lane with rel <= k
                const int32_t off = k - __shfl(rel, lo, 64);
                const int32_t of0 = __shfl(f0, lo, 64), olen = __shfl(len, lo, 64);
                const float ow = __shfl(w, lo, 64);
                if (k < total) {
                    seg[base + k] = make_int2(of0 + 64 * off, min(64, olen - 64 * off));
                    seg_w[base + k] = ow;
                }
            }
        }
    }
}

__global__ void k_seg_fill(const int32_t* __restrict__ csc_slot, const float* __restrict__ csc_w, const int32_t* __restrict__ chunk_off,
                           int32_t slot_base, int32_t q0, int32_t nq, int32_t nch, const int32_t* __restrict__ ptr,
                           int2* __restrict__ seg, float* __restrict__ seg_w, Half H) {
    seg_fill_body(csc_slot, csc_w, chunk_off, slot_base, q0, nq, nch, ptr, seg, seg_w, H);
}
__global__ void k_seg_fill_multi(const SegDesc* __restrict__ D, const int32_t* __restrict__ csc_slot, const float* __restrict__ csc_w,
                                 const int32_t* __restrict__ co_all, const int32_t* __restrict__ ptr_all, int2* __restrict__ seg, float* __restrict__ seg_w,
                                 const int32_t* __restrict__ row_of_entry, const int32_t* __restrict__ csr_idx, int32_t* __restrict__ start_all) {
    const SegDesc d = D[blockIdx.y];
    const Half H{(d.half || d.lim_chunks > 0) ? row_of_entry : nullptr, csr_idx, d.CH, start_all + d.q0, nullptr, d.use_only_rows ? row_of_entry : nullptr, d.only_from,
                 d.half ? d.sym_rows : -1, d.lim_from, d.lim_chunks};
    seg_fill_body(csc_slot, csc_w, co_all + d.co_off, d.slot_base, d.q0, d.nq, d.nch, ptr_all + d.cnt_off, seg, seg_w, H);
}
// chunk offsets / packed CSR / block-compressed tail CSR of all planned clusters: blockIdx.y = cluster of the plan
struct CoDesc {
    int32_t slot_base, n_slots, CH, nch, f0, f1, p_eff, has_tail;
    int64_t co_off, co_tail_off;
};
__global__ void k_chunk_offsets_multi(const CoDesc* __restrict__ D, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr_idx,
                                      int32_t* __restrict__ co_all) {
    const CoDesc d = D[blockIdx.y];
    chunk_offsets_body(rowptr, csr_idx, d.slot_base, d.n_slots, d.CH, d.nch, co_all + d.co_off);
}
__global__ void k_pack_csr_multi(const CoDesc* __restrict__ D, const int32_t* __restrict__ csr_idx, const float* __restrict__ csr_r, uint32_t* __restrict__ pk) {
    const CoDesc d = D[blockIdx.y];
    pack_csr_body(d.f0, d.f1, d.CH, csr_idx, csr_r, pk);
}
__global__ __launch_bounds__(256) void k_tail_blocks_multi(const CoDesc* __restrict__ D, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr_idx,
                                                           const float* __restrict__ csr_r, uint32_t* __restrict__ y_pk, int32_t* __restrict__ co_tail_all) {
    const CoDesc d = D[blockIdx.y];
    if (!d.has_tail) return;       // block-uniform
    tail_blocks_body(d.slot_base, d.n_slots, d.p_eff, rowptr, csr_idx, csr_r, y_pk, co_tail_all + d.co_tail_off);
}
__global__ void k_csr_samples(int64_t n_samples, const int32_t* __restrict__ csr_idx, int32_t* __restrict__ samp) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n_samples; k += (int64_t)gridDim.x * blockDim.x) samp[k] = csr_idx[k << FY_SAMPLE_SHIFT];
}
// every 64th entry of csr_idx (what the symmetric cut of the segment tables searches, Half::samp)
void build_csr_samples(Context* ctx, int64_t nnz, const int32_t* csr_idx, DevBuf<int32_t>& samp, hipStream_t st) {
    const int64_t n = (nnz + (1 << FY_SAMPLE_SHIFT) - 1) >> FY_SAMPLE_SHIFT;
    samp.alloc(ctx, (size_t)n + 1);
    if (n) k_csr_samples<<<grid_for(n), 256, 0, st ? st : ctx->stream>>>(n, csr_idx, samp.get());
    FY_KERNEL_CHECK();
}
void build_segments(Context* ctx, const int32_t* csc_slot, const float* csc_w, const int32_t* chunk_off, int32_t slot_base,
                    int32_t q0, int32_t nq, int32_t nch, SegTable& out, hipStream_t st, const int32_t* half_row_of_entry,
                    const int32_t* csr_idx, int32_t CH, const int32_t* only_rows_of_entry, int32_t only_rows_from, int64_t max_segments,
                    const int32_t* samples) {
    if (half_row_of_entry && !samples) FY_FAIL(FY_ERR_STATE, "internal: the symmetric segment table needs the CSR samples");
    if (!st) st = ctx->stream;
    const size_t np = (size_t)nch * ((size_t)nq + 1);
    out.ptr.alloc(ctx, np);
    out.cnt.alloc(ctx, np);
    out.scratch.alloc(ctx, half_row_of_entry ? (size_t)nq + 1 : 1);
    const Half H{half_row_of_entry, csr_idx, CH, out.scratch.get(), samples, only_rows_of_entry, only_rows_from};
    k_seg_counts<<<grid_for((int64_t)nq + 1), 256, 0, st>>>(csc_slot, chunk_off, slot_base, q0, nq, nch, out.cnt.get(), H);
    FY_KERNEL_CHECK();
    exclusive_scan_i32(ctx, out.cnt.get(), out.ptr.get(), np, st);
    int64_t total = max_segments;
    if (max_segments <= 0) {
        int32_t t32 = 0;   // the last count is 0 by construction: the last prefix is the total
        FY_HIP(hipMemcpyAsync(&t32, out.ptr.get() + (np - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
        FY_HIP(hipStreamSynchronize(st));
        total = t32;
    }
    out.n_seg = total;
    out.seg.alloc(ctx, (size_t)total);
    out.w.alloc(ctx, (size_t)total);
    if ((int64_t)nch * nq > 0) {
        k_seg_fill<<<grid_for((((int64_t)nq + 63) >> 6) * nch * 64, 256, 256 * 64), 256, 0, st>>>(csc_slot, csc_w, chunk_off, slot_base, q0, nq, nch, out.ptr.get(),
                                                                 out.seg.get(), out.w.get(), H);
        FY_KERNEL_CHECK();
    }
    if (max_segments <= 0) FY_HIP(hipStreamSynchronize(st));     // (the callers of the exact mode release their inputs right after the call)
}

// ================================================================ G build: co-rating row kernel + RM2 epilogue
// G[j][i] = (1-l)^2 (X^T X)_ji, the pure co-rating Gram (the rank-one part l (1-l) p_j b_i of the reference's inner sum is
// applied by the scoring kernels): symmetric, exactly zero for never co-rated pairs.
// Packed matrix format (clusters with >= pack24_min_items items): 3 bytes per element = 7 exponent + 17 mantissa bits of the fp32
// value, no sign (G >= 0) and no top exponent bit: every stored value is < 2.  The cluster's matrix is stored SCALED by
// 2^-c, c from an upper bound of its largest entry (rm2_score: cluster_scale); the scoring kernels never undo the scale -- q_j
// and e_uj are stored scaled by the same 2^-c, so every term G + q b + a e comes out scaled, its log2 is short by c, and the
// user's pvpi carries + n c ln 2.  Round 2 stored e8m16 (one mantissa bit less): the worst relative error of a score against the
// fp64 definition was 7.3e-6 of north_star's 1e-5; the 17th bit halves the rounding of every matrix element at the same 3 bytes.
constexpr int FY_P24_SHIFT = 6;      // float bits = packed << 6
struct MEpilogue {
    float* __restrict__ M;
    int64_t ldm;
    float w2;     // (1-l)^2
    double fx_inv;   // fixed-point accumulators: (1-l)^2 / fx_scale
    int pack24;   // 3 bytes per element: 8 exponent + 16 mantissa bits of the (non-negative) fp32, rounded to nearest
    float* __restrict__ Bmax;   // optional [row][ldb] in the 24-bit packed format (3 bytes per entry): maximum of the (rounded) row
                                // over every 256-column block
    int64_t ldb;
    int local_rows;   // 1: M / Bmax hold only this launch's rows, k-th row of the launch at index k (cooperative ranks)
    // column-panel mode (fy_rm2_kernels.hpp, "many clusters"): only the first panel_cols columns of a row are stored (row pitch
    // panel_cols), and the block maxima are taken over 64-column sub-blocks of the WHOLE row
    int32_t panel_cols;          // 0 = off
    float* __restrict__ Bmax64;  // [row][ldb64], 3 bytes per entry
    int64_t ldb64;
    // [row][ldb64]: (second largest value of the sub-block, 24 bits) << 8 | column of the largest inside the sub-block; written with
    // every non-zero Bmax64 entry.  k_bound_repair: a user who rated the column of the maximum is bounded by the second value
    uint32_t* __restrict__ Brep;
    // tail-bound launches of panel mode (k_tail_blocks below): the "matrix" is a column range of Bmax64 -- ldm columns are
    // written per row, the rows are `pitch` elements apart (0: the pitch is the row length) and the 24-bit rounding goes up
    int64_t pitch;
    int ceil24;
    // UNROUNDED fp64 copies of the rows the refinement pass reads (k_refine_rows, round 4; the fixed-point sums are exact, so these are
    // the matrix elements to 2^-53): the first head_rows rows of the matrix, whole, at
    // pitch ld_head (symmetric walks: only the columns behind the row are meaningful), and -- symmetric panel mode, whose head rows
    // are not walked over the tail rows' columns -- the first 256 columns of every row from tail_from on.  nullptr = off.
    double* __restrict__ head32;      // (fp64 since the first measurement: fp32 values left 4e-5 on a forced-small row with |score| = 0.15)
    int64_t ld_head;
    int32_t head_rows;
    double* __restrict__ tail32;
    int32_t tail_from;
};

// Epilogue of one (row, chunk) item of the RM2 row kernel: G[i][chunk] = w2 * acc -> 24-bit pack (or fp32), the block maxima, and the
// accumulators re-zeroed in the same pass.  All threads of the workgroup; `id` = (row of the launch << 8) | chunk.
// the 24-bit branch of the epilogue; COPY: this item also stores the unrounded fp64 images of its values for the refinement pass
// (MEpilogue::head32 / tail32) -- a separate instantiation, so that the items that do not (all but the first few hundred rows) run
// the code of round 3 (with the stores in the one loop the row kernel took 8.3 instead of 8.1 ms)
template <class ACC, bool COPY>
__device__ __forceinline__ void cooc_rm2_epilogue_pack(const CoocArgs& A, const MEpilogue& E, ACC* __restrict__ acc, int row, int mrow, int c0, int c1, int cb,
                                                       int first_bmax_block) {
        // four columns -> three dwords (c0 and c1 are multiples of 64)
        const int64_t row_cols = E.pitch ? E.pitch : (E.panel_cols ? E.panel_cols : E.ldm);
        const uint32_t radd = E.ceil24 ? (1u << FY_P24_SHIFT) - 1u : 1u << (FY_P24_SHIFT - 1);
        uint32_t* __restrict__ out3 = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(E.M) + (int64_t)mrow * row_cols * 3);
        for (int c4 = (cb >> 2) + threadIdx.x; 4 * c4 < c1; c4 += blockDim.x) {
            uint32_t v[4];
            // plane q of the accumulators holds the columns = q mod 4 (CoocArgs::acc_quarter): consecutive lanes, consecutive words
            const int qz = A.acc_quarter;
            ACC* ap = acc + (qz ? c4 - (c0 >> 2) : 4 * c4 - c0);
            const int qs = qz ? qz : 1;                       // stride between the four columns of the lane
            double f4[COPY ? 4 : 1];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float f;                                      // columns >= Ic were never touched: 0
                if constexpr (std::is_integral<ACC>::value) { const double d = (double)ap[q * qs] * E.fx_inv; f = (float)d; if constexpr (COPY) f4[q] = d; }
                else { f = E.w2 * (float)ap[q * qs]; if constexpr (COPY) f4[q] = (double)E.w2 * (double)ap[q * qs]; }
                ap[q * qs] = (ACC)0;
                v[q] = (__float_as_uint(f) + radd) >> FY_P24_SHIFT;    // 0 <= f < 2: 7 exponent + 17 mantissa bits, round to nearest (or up)
            }
            // the unrounded values (fp64 images of the fixed-point sums) the refinement pass re-scores ill-conditioned list rows from
            if constexpr (COPY) {
            if (E.head32 && mrow < E.head_rows && 4 * c4 < E.ld_head) {
                double2* hp = reinterpret_cast<double2*>(E.head32 + (int64_t)mrow * E.ld_head + 4 * c4);
                hp[0] = make_double2(f4[0], f4[1]);
                hp[1] = make_double2(f4[2], f4[3]);
            }
            if (E.tail32 && row >= E.tail_from && 4 * c4 < E.head_rows) {
                double2* tp = reinterpret_cast<double2*>(E.tail32 + (int64_t)(row - E.tail_from) * E.head_rows + 4 * c4);
                tp[0] = make_double2(f4[0], f4[1]);
                tp[1] = make_double2(f4[2], f4[3]);
            }
            }
            if (!E.panel_cols || 4 * c4 < E.panel_cols) {
                out3[3 * c4 + 0] = v[0] | (v[1] << 24);
                out3[3 * c4 + 1] = (v[1] >> 8) | (v[2] << 16);
                out3[3 * c4 + 2] = (v[2] >> 16) | (v[3] << 8);
            }
            if (E.Bmax64) {
                // 16 lanes = one 64-column sub-block.  Keys = value << 6 | column inside the sub-block (all different): the two
                // largest keys of the 64, by a four-step butterfly of (first, second) pairs
                const uint32_t cq = (uint32_t)((4 * c4) & 63);
                const uint32_t k0 = (v[0] << 6) | cq, k1 = (v[1] << 6) | (cq + 1), k2 = (v[2] << 6) | (cq + 2), k3 = (v[3] << 6) | (cq + 3);
                const uint32_t a1 = max(k0, k1), a2 = min(k0, k1), b1 = max(k2, k3), b2 = min(k2, k3);
                uint32_t t1 = max(a1, b1), t2 = max(min(a1, b1), max(a2, b2));
                // rotations inside the row of 16 lanes (DPP row_ror: no LDS instruction -- as ds_bpermute shuffles these were a third
                // of the kernel's LDS instructions in a 50-cluster job): after ror 8, 4, 2, 1 every lane has seen all 16, each once
#define FY_ROR16(x, n) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x120 + (n), 0xF, 0xF, false))
#define FY_TOP2_STEP(n)                                         \
    {                                                           \
        const uint32_t p1 = FY_ROR16(t1, n), p2 = FY_ROR16(t2, n); \
        t2 = max(min(t1, p1), max(t2, p2));                     \
        t1 = max(t1, p1);                                       \
    }
                FY_TOP2_STEP(8) FY_TOP2_STEP(4) FY_TOP2_STEP(2) FY_TOP2_STEP(1)
#undef FY_TOP2_STEP
#undef FY_ROR16
                const uint32_t m = t1 >> 6;
                if ((threadIdx.x & 15) == 0 && m) {
                    uint8_t* bp = reinterpret_cast<uint8_t*>(E.Bmax64) + ((int64_t)mrow * E.ldb64 + (c4 >> 4)) * 3;
                    bp[0] = (uint8_t)m;
                    bp[1] = (uint8_t)(m >> 8);
                    bp[2] = (uint8_t)(m >> 16);
                    if (E.Brep) E.Brep[(int64_t)mrow * E.ldb64 + (c4 >> 4)] = ((t2 >> 6) << 8) | (t1 & 63u);
                }
            }
            if (E.Bmax) {
                // maximum of the values exactly as the scoring kernel will unpack them; one wave = one 256-column
                // block (c0 and c1 are multiples of 256 when the bound matrix is requested, see pick_chunks)
                uint32_t m = max(max(v[0], v[1]), max(v[2], v[3]));     // non-negative floats order like their bit patterns
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
                if ((threadIdx.x & 63) == 0 && m && (c4 >> 6) >= first_bmax_block) {
                    // the maximum of 24-bit values is itself one: Bmax is stored in the same packed format (3 bytes per
                    // block, byte stores: the four blocks of a packed group belong to different waves or chunks), so
                    // the bound pass streams 768 instead of 1024 bytes per rated item.  (Zero maxima are not stored:
                    // the matrix was cleared before the launch.)
                    uint8_t* bp = reinterpret_cast<uint8_t*>(E.Bmax) + ((int64_t)mrow * E.ldb + (c4 >> 6)) * 3;
                    bp[0] = (uint8_t)m;
                    bp[1] = (uint8_t)(m >> 8);
                    bp[2] = (uint8_t)(m >> 16);
                }
            }
        }
}

template <class ACC>
__device__ __forceinline__ void cooc_rm2_epilogue(const CoocArgs& A, const MEpilogue& E, ACC* __restrict__ acc, int id) {
    const int lrow = id >> 8;
    const int row = A.row0 + lrow * (A.row_stride ? A.row_stride : 1);
    const int mrow = E.local_rows ? lrow : row;
    const int ch = id & 255;
    const int c0 = ch * A.CH;
    // the last chunk also writes the padding columns [Ic, ldm) so the scoring kernel may load whole 256-wide chunks
    const int c1 = (ch == A.nch - 1) ? (int)E.ldm : min(c0 + A.CH, (int)E.ldm);
    // symmetric walk: in the row's own chunk nothing in front of its 256-column diagonal block was accumulated (and the
    // mirror pass writes that part of the row); the block maxima of the diagonal block are the mirror pass's too
    const int cb = (cooc_row_is_cut(A, row) && c0 <= row) ? (row & ~255) : c0;
    const int first_bmax_block = A.half ? (row >> 8) + 1 : 0;
    if (E.pack24) {
        // (block-uniform) does this item hold values the refinement pass wants?
        const bool copy = E.head32 && (mrow < E.head_rows || (E.tail32 && row >= E.tail_from && c0 < E.head_rows));
        if (copy) cooc_rm2_epilogue_pack<ACC, true>(A, E, acc, row, mrow, c0, c1, cb, first_bmax_block);
        else cooc_rm2_epilogue_pack<ACC, false>(A, E, acc, row, mrow, c0, c1, cb, first_bmax_block);
    } else {
        float* __restrict__ out = E.M + (int64_t)mrow * E.ldm;
        for (int col = cb + threadIdx.x; col < c1; col += blockDim.x) {
            const int at = cooc_acc_index(col - c0, A.acc_quarter);
            if constexpr (std::is_integral<ACC>::value) out[col] = (float)((double)acc[at] * E.fx_inv);
            else out[col] = E.w2 * (float)acc[at];
            acc[at] = (ACC)0;
        }
    }
}

// Persistent workgroups pull (row, chunk) items from a global counter (rows are in popularity order: heavy items first);
// the epilogue re-zeroes the accumulators it reads, so an item costs one accumulate phase, one barrier, one epilogue and
// one barrier -- no dispatch, no separate clearing pass.
//
// Round 2: (1) accumulators in fp32 (ds_add_f32, ACC = float): half the LDS per column, so TWO workgroups share a CU and
// one streams rater slices while the other sits in its barrier / epilogue (one 16-wave workgroup per CU spent half of
// every item's 25 us waiting: three dependent round trips in front of the first slice load, then the epilogue); ACC =
// double keeps the round-1 arithmetic (FY_COOC_F64=1).  (2) The item counter is read TWO items ahead by thread 0, which
// also fetches the segment range of the next item while the current one is accumulated; after the barrier every wave issues
// the load of its first 64 segment descriptors of the NEXT item, and only then runs the epilogue: of the three round
// trips in front of an item's first slice load none is exposed any more.  (3) The epilogue no longer reads b and p.
// ROLE only names the instantiation (kernel traces): 0 = matrix rows, 1 = the tail rows' block bounds of panel mode (k_tail_blocks)
template <bool PK, class ACC>
__device__ __forceinline__ void cooc_rm2_body(const CoocArgs& A, const MEpilogue& E, int n_items, int* __restrict__ next_item) {
    ACC* __restrict__ acc = reinterpret_cast<ACC*>(fy_cooc_acc);
    __shared__ int sh_item, sh_s0, sh_s1, sh_id;
    const int CHp = cooc_lds_columns(A.CH);   // allocated (and zeroed) columns: the epilogue reads whole 256-column blocks
    for (int t = threadIdx.x; t < CHp; t += blockDim.x) acc[t] = (ACC)0;
    // items are handed out by a global counter: the chunks of a row differ a lot in weight (chunk 0 holds the popular
    // columns), a static stride would leave three quarters of the workgroups idle behind the chunk-0 owners
    // The counter hands out ITEM_GRAB consecutive items per atomic: one address takes ~82 M device-scope atomics per second on
    // this part (every launch of this kernel over short items ran at exactly that item rate, whatever the chunk width or the
    // walk: 99 k items in 1.20 ms, 46 k in 0.57 ms, 364 k in 4.4 ms), and in a cluster of 3 000 users an item is ~60 segments.
    // The base of the batch after the current one is fetched when a batch is started: its latency never shows.
    // Rows with thousands of segments (one cluster of 162 000 users) keep one item per atomic: eight consecutive items are the
    // chunks of ONE row there, and a batch of them unbalances the tail (7.9 -> 12.7 ms).
    const int ITEM_GRAB = A.item_grab > 0 ? A.item_grab : 1;
    int next = 0, next2 = 0;      // thread 0 only: the next item and the one after it
    int g_cur = 0, g_end = 0, g_next = 0;
    auto grab = [&]() __attribute__((always_inline)) {
        if (g_cur == g_end) {
            g_cur = g_next;
            g_end = g_cur + ITEM_GRAB;
            g_next = atomicAdd(next_item, ITEM_GRAB);
        }
        return g_cur++;
    };
    if (threadIdx.x == 0) {
        g_cur = atomicAdd(next_item, ITEM_GRAB);
        g_end = g_cur + ITEM_GRAB;
        g_next = atomicAdd(next_item, ITEM_GRAB);
        const int first = grab();
        next = grab();
        const int2 se = first < n_items ? A.item_seg[first] : make_int2(0, 0);
        sh_item = first;
        sh_s0 = se.x;
        sh_s1 = se.y;
        sh_id = first < n_items ? A.item_id[first] : 0;
    }
    __syncthreads();
    int item = sh_item, s0 = sh_s0, s1 = sh_s1, id = sh_id;
    // Every wave must have read the first item before thread 0 publishes the second one (it does so as soon as ITS share of
    // the first item is accumulated).  Inside the loop the barrier behind the epilogue separates the two; here nothing did:
    // with other kernels competing for the CU a delayed wave read the second item as its first -- thousands of wrong matrix
    // rows per job, only with several job lanes in flight (tools/debug_margin.py found it).
    __syncthreads();
    SegBatch batch = cooc_first_batch(A, s0, s1);
    while (item < n_items) {     // block-uniform
        int2 se_next = make_int2(0, 0);
        int id_next = 0;
        if (threadIdx.x == 0) {
            next2 = grab();
            if (next < n_items) { se_next = A.item_seg[next]; id_next = A.item_id[next]; }   // address known since the previous item
        }
        const int c0 = (id & 255) * A.CH;
        // symmetric walk: in the row's OWN chunk the slices start at a 64-aligned position in front of the row's column (segment
        // table: Half::start); the entries at or before it are masked here -- row i accumulates only the columns j > i
        const int row_of_item = A.row0 + (id >> 8) * (A.row_stride ? A.row_stride : 1);
        const int rmask = (cooc_row_is_cut(A, row_of_item) && row_of_item >= c0 && row_of_item < c0 + A.CH) ? row_of_item - c0 : -1;
        if constexpr (PK) cooc_accumulate_pk<ACC>(A, acc, s0, s1, batch, rmask);
        else cooc_accumulate_segments<false, ACC>(A, acc, s0, s1, c0, batch, rmask);
        // (every thread read sh_* of THIS item before the barrier that ended the previous epilogue)
        if (threadIdx.x == 0) { sh_item = next; sh_s0 = se_next.x; sh_s1 = se_next.y; sh_id = id_next; }
        __syncthreads();     // all atomics of this item are done; the next item is published
        const int nitem = sh_item;
        s0 = sh_s0;
        s1 = sh_s1;
        const int nid = sh_id;
        batch = cooc_first_batch(A, s0, s1);     // in flight during the epilogue
        cooc_rm2_epilogue<ACC>(A, E, acc, id);
        item = nitem;
        id = nid;
        if (threadIdx.x == 0) next = next2;
        __syncthreads();   // the accumulators are clean again before the next item's atomics
    }
}
template <bool PK, class ACC, int ROLE = 0>
__global__ void k_cooc_rm2(CoocArgs A, MEpilogue E, int n_items, int* __restrict__ next_item) {
    cooc_rm2_body<PK, ACC>(A, E, n_items, next_item);
}
// The matrix builds of SEVERAL clusters in one launch (panel mode, many clusters): blockIdx.y = cluster; its arguments and its item
// counter come from device memory.  The workgroups of cluster c + 1 start on the CUs that cluster c's workgroups leave: no launch
// gap and no idle tail between two clusters' row kernels (50 clusters at ML-25M shape: 100 launches back to back were 41 ms).
struct CoocLaunch {
    CoocArgs A;
    MEpilogue E;
    int32_t n_items, pad;
};
template <bool PK, class ACC, int ROLE = 0>
__global__ void k_cooc_rm2_multi(const CoocLaunch* __restrict__ D, int* __restrict__ counters) {
    const CoocLaunch& d = D[blockIdx.y];
    const CoocArgs A = d.A;
    const MEpilogue E = d.E;
    cooc_rm2_body<PK, ACC>(A, E, d.n_items, counters + blockIdx.y);
}
// the item lists of all those launches (CoocArgs::item_seg / item_id name the lists to fill)
__global__ void k_item_list_multi(const CoocLaunch* __restrict__ D) {
    const CoocArgs A = D[blockIdx.y].A;
    item_list_body(A, const_cast<int2*>(A.item_seg), const_cast<int32_t*>(A.item_id));
}

// ================================================================ mirror pass of the symmetric (half) walk
// After k_cooc_rm2 with CoocArgs::half, row i of the packed matrix holds the columns j >= 256 * (i / 256) (exact for j > i,
// zero elsewhere in the diagonal block) and Bmax holds the blocks behind the diagonal block.  G is symmetric, so the rest is
// a transposition: 24-bit elements, 256 x 128 source tiles through LDS (k_mirror_tiles: reads 256 row segments of 384 B,
// writes 128 row segments of 768 B -- one whole 256-column block of the destination rows, whose maximum is the missing
// Bmax entry), and the 256 x 256 diagonal blocks thread by thread (k_mirror_diag: 1 / 231 of the matrix).
__device__ __forceinline__ void fy_unpack24_raw(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t* v) {
    v[0] = d0 & 0xFFFFFFu;
    v[1] = (d0 >> 24) | ((d1 & 0xFFFFu) << 8);
    v[2] = (d1 >> 16) | ((d2 & 0xFFu) << 16);
    v[3] = d2 >> 8;
}
__device__ __forceinline__ uint32_t fy_load24(const uint32_t* __restrict__ row3, int e) {   // element e of a packed row
    const int g = e >> 2;
    uint32_t v[4];
    fy_unpack24_raw(row3[3 * g], row3[3 * g + 1], row3[3 * g + 2], v);
    return v[e & 3];
}
constexpr int MIRROR_PITCH = 772;    // bytes per LDS row: 768 + 4 (an odd number of dwords spreads the rows over the banks)

extern __shared__ __attribute__((aligned(16))) unsigned char fy_mirror_lds[];   // [128][MIRROR_PITCH]: the destination rows, packed
// `need` / `write_below` (round 4, the LAZY mirror of the pruned one-cluster job): the transposed tile is stored only for the column blocks
// somebody reads -- B < write_below (the seed columns) or need[B] != 0 (blocks with survivors, flagged on the device by
// k_flag_surviving_blocks) -- while the block maxima (Bmax_ != nullptr) are taken from every tile.  The full mirror is write_below =
// INT_MAX.  Measured: the survivors of the headline job (22 600 (user, block) pairs) lie in 8 of its 231 column blocks.
__device__ __forceinline__ void mirror_tiles_body(float* __restrict__ M_, int64_t ldm, int32_t Ic, float* __restrict__ Bmax_, int64_t ldb,
                                                  const int32_t* __restrict__ need = nullptr, int32_t write_below = 0x7FFFFFFF) {
    __shared__ uint32_t rowmax[128];
    const int tj = blockIdx.x, B = blockIdx.y;
    if (tj < 2 * (B + 1)) return;                         // only tiles strictly behind the diagonal block are sources
    const int dst_row0 = 128 * tj;
    if (dst_row0 >= Ic) return;                           // destination rows are real rows (the padding columns have none)
    const bool write = B < write_below || (need && need[B] != 0);      // block-uniform
    if (!write && !Bmax_) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int64_t pitch = ldm * 3;
    unsigned char* __restrict__ Mb = reinterpret_cast<unsigned char*>(M_);
    if (threadIdx.x < 128) rowmax[threadIdx.x] = 0;
    __syncthreads();
    // ---- load: a wave takes two source rows per step (32 lanes x 12 bytes = 128 elements each)
    const int m = lane & 31, h = lane >> 5;
    uint32_t mx[4] = {0, 0, 0, 0};
    for (int r = 2 * wave + h; r < 256; r += 2 * nwaves) {
        const int src_row = 256 * B + r;
        uint32_t v[4] = {0, 0, 0, 0};
        if (src_row < Ic) {
            const uint32_t* __restrict__ p = reinterpret_cast<const uint32_t*>(Mb + (int64_t)src_row * pitch + (int64_t)dst_row0 * 3) + 3 * m;
            fy_unpack24_raw(p[0], p[1], p[2], v);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (write) {
                unsigned char* d = fy_mirror_lds + (4 * m + q) * MIRROR_PITCH + 3 * r;    // destination row 4m + q, element r
                d[0] = (unsigned char)v[q];
                d[1] = (unsigned char)(v[q] >> 8);
                d[2] = (unsigned char)(v[q] >> 16);
            }
            mx[q] = max(mx[q], v[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (mx[q]) atomicMax(&rowmax[4 * m + q], mx[q]);
    __syncthreads();
    // ---- store: one destination row (768 bytes = one 256-column block) per wave step
    for (int c = wave; write && c < 128; c += nwaves) {
        const int dst_row = dst_row0 + c;
        if (dst_row >= Ic) break;
        const uint32_t* __restrict__ sp = reinterpret_cast<const uint32_t*>(fy_mirror_lds + c * MIRROR_PITCH) + 3 * lane;
        uint32_t* __restrict__ dp = reinterpret_cast<uint32_t*>(Mb + (int64_t)dst_row * pitch + (int64_t)256 * B * 3) + 3 * lane;
        dp[0] = sp[0];
        dp[1] = sp[1];
        dp[2] = sp[2];
    }
    if (Bmax_ && threadIdx.x < 128 && dst_row0 + (int)threadIdx.x < Ic) {
        const uint32_t pv = rowmax[threadIdx.x];
        uint8_t* bp = reinterpret_cast<uint8_t*>(Bmax_) + ((int64_t)(dst_row0 + threadIdx.x) * ldb + B) * 3;
        bp[0] = (uint8_t)pv;
        bp[1] = (uint8_t)(pv >> 8);
        bp[2] = (uint8_t)(pv >> 16);
    }
}

__global__ __launch_bounds__(1024) void k_mirror_tiles(float* __restrict__ M_, int64_t ldm, int32_t Ic, float* __restrict__ Bmax_, int64_t ldb,
                                                       const int32_t* __restrict__ need, int32_t write_below) {
    mirror_tiles_body(M_, ldm, Ic, Bmax_, ldb, need, write_below);
}
// Block maxima alone (lazy mirror, first pass, the column blocks nobody reads below the diagonal): Bmax[j][B] = max over the rows i of block B
// of G[i][j] for the 256 columns j of a tile strictly behind block B.  A wave reads WHOLE 768-byte row segments (one 12-byte load per
// lane, 16 rows in flight per workgroup step) -- the transposing tile kernel reads 384-byte half segments and stages them through LDS,
// which this pass has no use for; the 16 waves' maxima meet in LDS and each thread stores one 3-byte entry.
__global__ __launch_bounds__(1024) void k_colmax_upper(const float* __restrict__ M_, int64_t ldm, int32_t Ic, float* __restrict__ Bmax_, int64_t ldb, int32_t first_block) {
    __shared__ uint32_t colmax[256];
    const int tj = blockIdx.x, B = first_block + blockIdx.y;      // tile = columns [256 tj, 256 tj + 256) x rows of block B
    if (tj <= B) return;                                           // strictly behind the diagonal block
    const int col0 = 256 * tj;
    if (col0 >= Ic) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pitch = ldm * 3;
    const unsigned char* __restrict__ Mb = reinterpret_cast<const unsigned char*>(M_);
    if (threadIdx.x < 256) colmax[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mx[4] = {0, 0, 0, 0};
    const int r_end = min(256, Ic - 256 * B);
    uint32_t d[4][3];
#pragma unroll
    for (int step = 0; step < 4; step++) {                         // 16 waves x 16 rows: four loads in flight per lane
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int r = wave * 16 + step * 4 + x;
            const uint32_t* __restrict__ p = reinterpret_cast<const uint32_t*>(Mb + (int64_t)(256 * B + min(r, r_end - 1)) * pitch + (int64_t)col0 * 3) + 3 * lane;
            d[x][0] = p[0]; d[x][1] = p[1]; d[x][2] = p[2];
        }
#pragma unroll
        for (int x = 0; x < 4; x++) {
            uint32_t v[4];
            fy_unpack24_raw(d[x][0], d[x][1], d[x][2], v);
#pragma unroll
            for (int q = 0; q < 4; q++) mx[q] = max(mx[q], v[q]);      // (a clamped row repeats the block's last row: the maximum is unchanged)
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (mx[q]) atomicMax(&colmax[4 * lane + q], mx[q]);
    __syncthreads();
    if (threadIdx.x < 256 && col0 + (int)threadIdx.x < Ic) {
        const uint32_t pv = colmax[threadIdx.x];
        uint8_t* bp = reinterpret_cast<uint8_t*>(Bmax_) + ((int64_t)(col0 + threadIdx.x) * ldb + B) * 3;
        bp[0] = (uint8_t)pv;
        bp[1] = (uint8_t)(pv >> 8);
        bp[2] = (uint8_t)(pv >> 16);
    }
}
// need[b] = 1 for every block b some user of the batch keeps (the survivor lists of k_bound_select)
__global__ void k_flag_surviving_blocks(int32_t n_users, const int32_t* __restrict__ n_quads, const uint16_t* __restrict__ surv, int64_t ldb, int32_t* __restrict__ need) {
    for (int32_t u = blockIdx.x; u < n_users; u += gridDim.x)
        for (int k = threadIdx.x; k < n_quads[u]; k += blockDim.x) {
            const int b = surv[(int64_t)u * ldb + k];
            if (need[b] == 0) need[b] = 1;      // (benign race: every writer stores 1)
        }
}
// diagonal blocks: thread c owns row 256 B + c; element (c, r) for r < c is element (r, c) of a row above (a 3-byte gather),
// the rest of the row's segment is its own; the maximum over the completed segment is Bmax[row][B]
__device__ __forceinline__ void mirror_diag_body(float* __restrict__ M_, int64_t ldm, int32_t Ic, float* __restrict__ Bmax_, int64_t ldb) {
    const int B = blockIdx.x, c = threadIdx.x;
    const int row = 256 * B + c;
    if (row >= Ic) return;
    const int64_t pitch = ldm * 3;
    unsigned char* __restrict__ Mb = reinterpret_cast<unsigned char*>(M_);
    uint32_t* __restrict__ own = reinterpret_cast<uint32_t*>(Mb + (int64_t)row * pitch + (int64_t)256 * B * 3);
    uint32_t best = 0;
    for (int g = 0; g < 64; g++) {
        uint32_t v[4];
        if (4 * g >= c) {                                  // untouched part of the own row (columns >= c)
            fy_unpack24_raw(own[3 * g], own[3 * g + 1], own[3 * g + 2], v);
        } else {
            if (4 * g + 3 >= c) fy_unpack24_raw(own[3 * g], own[3 * g + 1], own[3 * g + 2], v);   // the group that straddles the diagonal
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = 4 * g + q;
                if (r < c) v[q] = fy_load24(reinterpret_cast<const uint32_t*>(Mb + (int64_t)(256 * B + r) * pitch + (int64_t)256 * B * 3), c);
            }
            own[3 * g + 0] = v[0] | (v[1] << 24);
            own[3 * g + 1] = (v[1] >> 8) | (v[2] << 16);
            own[3 * g + 2] = (v[2] >> 16) | (v[3] << 8);
        }
        best = max(max(best, max(v[0], v[1])), max(v[2], v[3]));
    }
    if (Bmax_) {
        uint8_t* bp = reinterpret_cast<uint8_t*>(Bmax_) + ((int64_t)row * ldb + B) * 3;
        bp[0] = (uint8_t)best;
        bp[1] = (uint8_t)(best >> 8);
        bp[2] = (uint8_t)(best >> 16);
    }
}

__global__ __launch_bounds__(256) void k_mirror_diag(float* __restrict__ M_, int64_t ldm, int32_t Ic, float* __restrict__ Bmax_, int64_t ldb) {
    mirror_diag_body(M_, ldm, Ic, Bmax_, ldb);
}
// the panels of all clusters of a symmetric panel-mode job: blockIdx.z (tiles, column maxima) / blockIdx.y (diagonal blocks) = cluster
struct PanelDesc {
    float* Gp;
    float* Bmax64;
    uint32_t* Brep;
    int64_t panel_cols, ldb64;
    int32_t Ic, p_eff, nsub, pad;
};
__global__ __launch_bounds__(1024) void k_mirror_tiles_multi(const PanelDesc* __restrict__ D) {
    const PanelDesc d = D[blockIdx.z];
    if ((int)blockIdx.x >= (int)(d.panel_cols / 128) || (int)blockIdx.y >= (int)(d.panel_cols / 256) - 1) return;
    mirror_tiles_body(d.Gp, d.panel_cols, d.p_eff, nullptr, 0);
}
__global__ __launch_bounds__(256) void k_mirror_diag_multi(const PanelDesc* __restrict__ D) {
    const PanelDesc d = D[blockIdx.y];
    if ((int)blockIdx.x >= (int)(d.panel_cols / 256)) return;
    mirror_diag_body(d.Gp, d.panel_cols, d.p_eff, nullptr, 0);
}
// Symmetric panel mode: the sub-block maxima of the HEAD rows (i < p_eff) from the stored panel, by symmetry:
//     Bmax64[i][sb] = max_{j in sub-block sb} G[i][j] = max_j Gp[j][i],      Brep[i][sb] = (second largest << 8) | (j of the largest - 64 sb)
// for every sub-block sb of the row -- the panel Gp holds column i of EVERY row j (head rows after the mirror pass, tail rows from
// their walk over the head chunks).  Work-group = 256 columns x 16 sub-blocks (1024 rows of the panel): wave w takes sub-blocks
// 4 w .. 4 w + 3, a lane four columns (12-byte loads, a wave reads 768 contiguous bytes of a row); the results go through LDS so
// that a thread writes the 16 entries of ONE row of Bmax64 / Brep (48 / 64 contiguous bytes).  Every entry of the head rows is
// written, zeros too: what the row kernel's epilogue left there (sub-blocks of the partly walked own chunk) is overwritten.
__global__ __launch_bounds__(256) void k_panel_colmax(const PanelDesc* __restrict__ D) {
    const PanelDesc dsc = D[blockIdx.z];
    const float* __restrict__ Gp_ = dsc.Gp;
    float* __restrict__ Bmax64_ = dsc.Bmax64;
    uint32_t* __restrict__ Brep = dsc.Brep;
    const int64_t panel_cols = dsc.panel_cols, ldb64 = dsc.ldb64;
    const int32_t Ic = dsc.Ic, p_eff = dsc.p_eff, nsub = dsc.nsub;
    if ((int)blockIdx.x * 256 >= p_eff || (int)blockIdx.y * 16 >= nsub) return;
    __shared__ uint32_t sh_m[16][257], sh_r[16][257];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col0 = blockIdx.x * 256, sbg = blockIdx.y * 16;
    const int c = col0 + 4 * lane;
    const unsigned char* __restrict__ base = reinterpret_cast<const unsigned char*>(Gp_) + (int64_t)c * 3;
    const int64_t pitch = panel_cols * 3;
    for (int s = 0; s < 4; s++) {
        const int sb = sbg + 4 * wave + s;
        uint32_t t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
        if (sb < nsub) {
            const int j0 = 64 * sb, nr = min(64, Ic - j0);
            for (int r = 0; r < nr; r += 8) {
                uint32_t d[8][3];
#pragma unroll
                for (int x = 0; x < 8; x++) {
                    const uint32_t* __restrict__ p = reinterpret_cast<const uint32_t*>(base + (int64_t)(j0 + min(r + x, nr - 1)) * pitch);
                    d[x][0] = p[0]; d[x][1] = p[1]; d[x][2] = p[2];
                }
#pragma unroll
                for (int x = 0; x < 8; x++) {
                    if (r + x < nr) {
                        uint32_t v[4];
                        fy_unpack24_raw(d[x][0], d[x][1], d[x][2], v);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t key = (v[q] << 6) | (uint32_t)(r + x);
                            t2[q] = max(min(t1[q], key), t2[q]);
                            t1[q] = max(t1[q], key);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            sh_m[4 * wave + s][4 * lane + q] = t1[q] >> 6;
            sh_r[4 * wave + s][4 * lane + q] = (t1[q] >> 6) ? (((t2[q] >> 6) << 8) | (t1[q] & 63u)) : 0u;
        }
    }
    __syncthreads();
    const int i = col0 + threadIdx.x;
    if (i < p_eff) {
        uint32_t m[16];
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] = sh_m[k][threadIdx.x];
        // 16 entries of 3 bytes = 12 dwords at a 16-byte aligned address (ldb64 and sbg are multiples of 16)
        uint32_t* __restrict__ bp = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(Bmax64_) + ((int64_t)i * ldb64 + sbg) * 3);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t* v = m + 4 * g;
            bp[3 * g + 0] = v[0] | (v[1] << 24);
            bp[3 * g + 1] = (v[1] >> 8) | (v[2] << 16);
            bp[3 * g + 2] = (v[2] >> 16) | (v[3] << 8);
        }
        uint32_t* __restrict__ rp = Brep + (int64_t)i * ldb64 + sbg;
#pragma unroll
        for (int k = 0; k < 16; k++) rp[k] = sh_r[k][threadIdx.x];
    }
}

// full mirror: write_below = INT_MAX.  Lazy mirror, first pass (block maxima from every tile, lower triangle only for the column blocks
// in front of write_below + the diagonal blocks): need = nullptr, write_below = seed blocks.  Second pass (after the survivors are
// known): Bmax = nullptr, need = the flags, write_below = 0, diag = false -- a tile of an unflagged block leaves at once.
static void launch_mirror(Context* ctx, float* M, int64_t ldm, int32_t Ic, float* Bmax, int64_t ldb, hipStream_t st, const int32_t* need = nullptr,
                          int32_t write_below = 0x7FFFFFFF, bool diag = true) {
    const int nblk = (int)(ldm / 256), ntile = (int)(ldm / 128);
    if (nblk > 1) {
        FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mirror_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * MIRROR_PITCH));
        // first pass of the lazy mirror: the transposing kernel only over the column blocks it writes (those in front of write_below),
        // the block maxima of all other blocks from the streaming kernel
        const bool split = !need && Bmax && write_below < nblk - 1;
        const int ny = split ? write_below : nblk - 1;
        if (ny > 0) {
            k_mirror_tiles<<<dim3(ntile, ny), 1024, 128 * MIRROR_PITCH, st>>>(M, ldm, Ic, Bmax, ldb, need, write_below);
            FY_KERNEL_CHECK();
        }
        if (split) {
            k_colmax_upper<<<dim3(nblk, nblk - 1 - write_below), 1024, 0, st>>>(M, ldm, Ic, Bmax, ldb, write_below);
            FY_KERNEL_CHECK();
        }
    }
    if (diag) {
        k_mirror_diag<<<nblk, 256, 0, st>>>(M, ldm, Ic, Bmax, ldb);
        FY_KERNEL_CHECK();
    }
}

static void cooc_rm2_allow_lds() {
    const int bytes = 160 * 1024 - 256;     // everything but the kernel's few static words
    FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_rm2<false, double>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_rm2<true, double>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_rm2<false, float>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_rm2<true, float>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_rm2<true, unsigned long long>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_rm2<true, uint32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_rm2<true, unsigned long long, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_rm2_multi<true, unsigned long long, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cooc_rm2_multi<true, unsigned long long, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
}
static void stray_allow_lds();

#include "fy_rm2_kernels.hpp"   // scoring, top-N, branch-and-bound and cooperative-rank kernels (part of this translation unit)

static void stray_allow_lds() {   // k_score_stray: (Uc + 1) rater offsets of dynamic LDS, up to 128 KB
    FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_score_stray), hipFuncAttributeMaxDynamicSharedMemorySize, (STRAY_UCAP + 1) * (int)sizeof(int32_t)));
}

void launch_topn_rows(Context* ctx, hipStream_t st, const float* S, int64_t ldS, int32_t n_cols, int32_t n_rows,
                      const int32_t* n_out, const int32_t* out_off, const int32_t* rank_item_raw, const int32_t* slot2du,
                      const int32_t* uid, int32_t slot0, int32_t aux_value, int32_t* out_user, int32_t* out_item,
                      float* out_score, int32_t* out_aux, int32_t* overflow, int32_t* any_overflow, int32_t top_n_hint) {
    if (n_rows <= 0) return;
    if (!st) st = ctx->stream;
    // n_out / out_off are indexed by (slot - slot_lo): pass slot_lo = slot0 so that row u reads entry u
    TopNArgs TA{S, ldS, n_cols, n_out, out_off, rank_item_raw, slot2du, uid, slot0, slot0, aux_value, out_user, out_item, out_score, out_aux,
                0, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr};
    FY_HIP(hipMemsetAsync(any_overflow, 0, sizeof(int32_t), st));
    if (top_n_hint > TOPN_LONG) k_topn_long<<<n_rows, 256, (size_t)fy_topn_long_cap(top_n_hint) * 8, st>>>(TA, overflow, any_overflow, 0, fy_topn_long_cap(top_n_hint));
    else k_topn_fast<<<n_rows, 256, 0, st>>>(TA, overflow, any_overflow, 0);
    FY_KERNEL_CHECK();
    k_topn_select<<<n_rows, 256, 0, st>>>(TA, overflow, any_overflow);
    FY_KERNEL_CHECK();
}

}  // namespace fy

// ================================================================ job orchestration
using namespace fy;

// Everything about a job that depends only on the RATINGS and the CLUSTERING (and the rank's share): the CSR / CSC of fy_prep,
// the statistics summed per (cluster, item), and -- filled by the first fy_rm2_score -- the row kernel's tables.  None of it
// depends on lambda, the list length or numberOfItems.  The reference rebuilds all of it in every job (its mappers re-read and
// re-shuffle the ratings, RM2Job.java:130-258); here it is kept on the fy_ratings object (fy_ratings::rm2_cache) and a later
// job over the same ratings and the same clustering starts from it ("warm": fy_stats::prepared_from_cache).  fy_rm2_params::flags
// & FY_RM2_NO_CACHE builds it afresh and does not keep it: the "cold" job, which bench.py times as its headline value.
struct TableCache {
    bool valid = false;
    int64_t total_segments = 0;         // segments in all tables of the job (fy_stats::cooc_segments)
    std::vector<int32_t> sig;           // what the tables were built for: per planned cluster (c, CH, nch, half, panel, p_eff, tail_chunks), + flags
    bool have_x = false;
    DevBuf<float> csc_x, csc_x_over_s;  // x = r / s_v (and x / s_v for the packed walk) per CSC entry: ratings only
    DevBuf<uint32_t> csr_pk, y_pk;      // packed CSR (chunk-relative columns), block-compressed CSR of the tail rows
    DevBuf<int32_t> csc_rank, samples;      // row of every CSC entry; every 64th CSR entry (symmetric cut)
    std::vector<SegTable> segs, segs_tail;
    // many clusters: ONE table over all of them (build_tables_all); segs / segs_tail are then views into these
    DevBuf<int32_t> g_co, g_co_tail, g_cnt, g_ptr, g_start;
    DevBuf<int2> g_seg;
    DevBuf<float> g_w;
};
struct RM2Static {
    // key
    int32_t K = 0, rank = 0, world = 1;
    int32_t max_item = -1;      // largest raw item id of the ratings (fy_ratings::max_item): the refinement pass maps raw ids to columns by table
    bool has_count = false;
    std::vector<int32_t> map_user, map_cluster, cluster_count;
    // fy_rm2_prepare
    Prepared P;
    int32_t slot_lo = 0, slot_hi = 0;
    DevBuf<double> partial;     // nI + 1 : this rank's exchange buffer
    // sharded prep (fy_prep.hpp: shard_ratings_by_cluster): P holds this rank's clusters alone and the exchange buffer is by raw id
    bool sharded = false;
    int64_t n_raw_users = 0, n_raw_items = 0;
    std::vector<int32_t> owner;         // [cluster] -> rank (world: empty cluster)
    DevBuf<double> partial_raw;         // n_raw_items + 1 + n_raw_users + 1
    int deferred_code = 0;              // a failure only this rank can see (a duplicate rating / clusteringCount mismatch inside an owned
    std::string deferred_msg;           // cluster): it travels with the statistics so that every rank fails instead of waiting for this one
    int64_t exchange_len() const { return sharded ? n_raw_items + 1 + n_raw_users + 1 : (int64_t)P.nI + 1; }
    // per (cluster, item) in rank order, from the one CSC walk of fy_rm2_prepare (k_pair_pass)
    DevBuf<double> b_rank, usum_slot;
    DevBuf<long long> walk_rank;
    DevBuf<int32_t> cnt_rank, deg_slot;
    DevBuf<float> fx_rank;              // 3 per (cluster, item): see PairPass
    std::vector<float> fx_bounds;       // 3 per cluster (host): max sum of weights, max weight, max rating
    double ms_build = 0;
    TableCache tables;
    bool matches(const fy_rm2_params* prm, int64_t n_map, const int32_t* mu, const int32_t* mc, const int32_t* cc) const {
        if (K != prm->number_of_clusters || rank != prm->rank || world != prm->world || (int64_t)map_user.size() != n_map || has_count != (cc != nullptr)) return false;
        if (n_map && (memcmp(map_user.data(), mu, (size_t)n_map * 4) || memcmp(map_cluster.data(), mc, (size_t)n_map * 4))) return false;
        if (cc && memcmp(cluster_count.data(), cc, (size_t)K * 4)) return false;
        return true;
    }
};

struct fy_rm2_job {
    Context* ctx = nullptr;
    fy_rm2_params prm{};
    std::shared_ptr<RM2Static> S;
    // (the static part under the names the job code has always used)
    Prepared& P;
    int32_t &slot_lo, &slot_hi;
    DevBuf<double>& partial;
    DevBuf<double> stats;       // nI + 1 : global sums (+ counter)
    DevBuf<double> stats_raw;   // sharded prep: the sum over ranks of the raw-id exchange buffers
    bool have_global = false;
    double ms_prepare = 0;
    bool from_cache = false;
    fy_collectives coll{};      // process-group collectives (optional)
    bool have_coll = false;
    DevBuf<double> gathered;    // world * (nI + 1): the statistics all-gathered by fy_rm2_score itself
    DevBuf<double>&b_rank, &usum_slot;
    DevBuf<long long>& walk_rank;
    DevBuf<int32_t>&cnt_rank, &deg_slot;
    DevBuf<float>& fx_rank;
    std::vector<float>& fx_bounds;
    bool count_balanced = false;   // scoring ownership of the users: equal counts instead of equal work (see owner_range)
    explicit fy_rm2_job(std::shared_ptr<RM2Static> s)
        : S(std::move(s)), P(S->P), slot_lo(S->slot_lo), slot_hi(S->slot_hi), partial(S->partial), b_rank(S->b_rank), usum_slot(S->usum_slot),
          walk_rank(S->walk_rank), cnt_rank(S->cnt_rank), deg_slot(S->deg_slot), fx_rank(S->fx_rank), fx_bounds(S->fx_bounds) {}
};

// Which users does rank k emit lists for?  By default the work-balanced slot range of fy_prep (the rank scores its
// users alone, so equal work).  A single cluster scored cooperatively shares all scoring work anyway and only the top-N
// merge is per owner: equal COUNTS keep the padded reduce-scatter segments small (slots are degree-descending, the
// work-balanced last rank would own four times the average number of users).
static void owner_range(const fy_rm2_job* J, int k, int32_t& lo, int32_t& hi) {
    if (J->S->sharded) { lo = 0; hi = k == J->prm.rank ? J->P.nU : 0; return; }      // this rank's structure holds its own clusters alone
    if (!J->count_balanced) { rank_slot_range(J->P, k, J->prm.world, lo, hi); return; }
    const int64_t n = J->P.nU, W = J->prm.world;
    lo = (int32_t)(n * k / W);
    hi = (int32_t)(n * (k + 1) / W);
}

// launch-shape knobs: fy::Tuning (fy_common.hpp), read from the environment ONCE when the context is created
// (fy_context_create / fy_context_reload_tuning in fy_api.hip) -- no job reads the environment.
using ScoreTune = fy::Tuning;
static inline const ScoreTune& score_tune(const Context* ctx) { return ctx->tune; }

// Exponent k of the fixed-point scale 2^k of a cluster (CoocArgs::fx_scale): the largest k with (largest contribution) * 2^k < 2^51
// and (largest possible Gram entry) * 2^k < 2^62, from the cluster's bounds (sum and maximum of the segment weights r / s^2 per
// item, largest rating).  Returns a negative number when the bounds are unusable (the fp64 path is taken then).
static int fx_exponent(const float* bounds3) {
    const double wsum = bounds3[0], wmax = bounds3[1], rmax = bounds3[2];
    if (!(wsum > 0.0) || !(wmax > 0.0) || !(rmax > 0.0) || !std::isfinite(wsum * rmax)) return -1;
    const int k1 = 51 - (std::ilogb(wmax * rmax) + 1), k2 = 62 - (std::ilogb(wsum * rmax) + 1);
    const int k = std::min(std::min(k1, k2), 1000);
    return k >= 24 ? k : -1;
}

// launch shape of the RM2 row kernel: as many workgroups per CU as the LDS accumulators allow (fp32: two for ML-25M's
// 19 712-column chunks), 2048 threads per CU at most
// items per atomic of the row kernel's work counter: 8 where an item is short (fewer than ~12 000 co-rating contributions, i.e.
// < 256 segments on average; `visits` = the cluster's sum of n_u^2, an upper bound of the launch's contributions), else 1
static int cooc_item_grab(int64_t visits, int64_t n_items) { return n_items > 0 && visits / n_items < 12000 ? 8 : 1; }

static void launch_cooc_rm2(Context* ctx, const ScoreTune& tune, bool use_pk, const CoocArgs& CA_, const MEpilogue& ME, int n_items,
                            int32_t* counter, hipStream_t st) {
    if (n_items <= 0) return;
    CoocArgs CA = CA_;
    CA.acc_quarter = tune.cooc_planes ? cooc_lds_columns(CA.CH) / 4 : 0;
    const bool acc32 = CA.acc32 && use_pk && CA.fx_scale > 0.0 && !tune.cooc_f32;
    const size_t lds = (size_t)cooc_lds_columns(CA.CH) * ((tune.cooc_f32 || acc32) ? 4 : 8);
    const int by_lds = (int)std::max<size_t>(1, (160 * 1024 - 512) / (lds + 64));
    int block = tune.cooc_block;
    if (!block) block = by_lds >= 4 ? 256 : (by_lds >= 2 ? 512 : 1024);
    const int per_cu = std::max(1, std::min(by_lds, 2048 / block));
    const int grid = std::min(n_items, ctx->num_cus * per_cu);
    if (acc32) {
        k_cooc_rm2<true, uint32_t><<<grid, block, lds, st>>>(CA, ME, n_items, counter);
    } else if (use_pk && CA.fx_scale > 0.0 && !tune.cooc_f32) {
        if (ME.ceil24) k_cooc_rm2<true, unsigned long long, 1><<<grid, block, lds, st>>>(CA, ME, n_items, counter);
        else k_cooc_rm2<true, unsigned long long><<<grid, block, lds, st>>>(CA, ME, n_items, counter);
    } else if (!tune.cooc_f32) {
        if (use_pk) k_cooc_rm2<true, double><<<grid, block, lds, st>>>(CA, ME, n_items, counter);
        else k_cooc_rm2<false, double><<<grid, block, lds, st>>>(CA, ME, n_items, counter);
    } else {
        if (use_pk) k_cooc_rm2<true, float><<<grid, block, lds, st>>>(CA, ME, n_items, counter);
        else k_cooc_rm2<false, float><<<grid, block, lds, st>>>(CA, ME, n_items, counter);
    }
    FY_KERNEL_CHECK();
}

// The upper triangle of the co-rating Gram of a ONE-cluster structure with caller-chosen weights, as fp32 (the item-item
// similarity build, fy_itemsim.hip): G[i][j] = sum_v w_vi r_vj for j > i, zero for (i & ~255) <= j <= i, rows `ldm` floats
// apart; nothing in front of a row's diagonal 256-column block is written.  The same symmetric walk, segment table, item
// list and fixed-point accumulators as the RM2 matrix build -- only the epilogue's scale differs (no (1 - lambda)^2).
// `csc_w` = the weight of every CSC entry (the row side of a product), the column side is the raw rating of the packed
// CSR; bounds3 = {largest column sum of the weights, largest weight, largest rating} (fx_exponent).  Returns false when the
// fixed-point scale cannot be used (the caller keeps its own path); everything is queued on the context's stream.
bool fy::gram_half_build(Context* ctx, const Prepared& P, const float* csc_w, const float* bounds3, float* G, int64_t ldm, double* ms_tables,
                         double* ms_walk) {
    const ScoreTune tune = score_tune(ctx);
    const int32_t Ic = P.nP;
    if (P.K != 1 || !P.ratings_fp16_exact || !P.ratings_positive || Ic <= 0) return false;
    int fxk = fx_exponent(bounds3);
    if (fxk < 0) return false;
    // Ratings that are multiples of 2^-m (half stars: m = 1): every product r r' 2^2m is an integer.  When the largest product
    // stays below 2^24 and the largest possible sum below 2^32 the walk accumulates in 32 bits (ds_add_u32: 6.5 cycles per wave
    // instruction at random addresses against 10.8 for ds_add_u64, profiles/r2/micro_lds_atomic_rate.txt) and a chunk holds
    // twice the columns: two column chunks instead of three at ML-25M shape.  Exact, like the 64-bit sums.
    const int m = P.ratings_frac_bits;
    const bool acc32 = tune.isim_acc32 && m <= 4 && (double)bounds3[2] * bounds3[2] * std::ldexp(1.0, 2 * m) < 16777216.0 &&
                       (double)bounds3[0] * bounds3[2] * std::ldexp(1.0, 2 * m) < 4294967296.0;
    if (acc32) fxk = 2 * m;
    int32_t CH, nch;
    pick_chunks(Ic, acc32 && !tune.cooc_max_ch_forced ? 2 * tune.cooc_max_ch : tune.cooc_max_ch, CH, nch);
    if (nch >= 256) return false;                      // (the item ids of the row kernel hold the chunk in 8 bits)
    hipStream_t st = ctx->stream;
    cooc_rm2_allow_lds();
    EventTimer t_tab(ctx), t_walk(ctx);
    const size_t s0 = t_tab.begin();
    DevBuf<uint32_t> csr_pk(ctx, (size_t)P.nnz);
    k_pack_csr<<<grid_for(P.nnz), 256, 0, st>>>(0, (int32_t)P.nnz, CH, P.csr_idx.get(), P.csr_r.get(), csr_pk.get());
    FY_KERNEL_CHECK();
    DevBuf<int32_t> co(ctx, (size_t)P.nU * (nch + 1)), csc_rank(ctx, (size_t)P.nnz);
    build_chunk_offsets(ctx, P.rowptr.get(), P.csr_idx.get(), 0, P.nU, CH, nch, co.get(), st);
    k_csc_rank<<<grid_for(P.nnz), 256, 0, st>>>(P.nnz, P.csc_pair.get(), P.pair_rank.get(), csc_rank.get());
    FY_KERNEL_CHECK();
    SegTable segs;
    DevBuf<int32_t> samples;
    build_csr_samples(ctx, P.nnz, P.csr_idx.get(), samples, st);
    build_segments(ctx, P.csc_slot.get(), csc_w, co.get(), 0, 0, (int32_t)P.nnz, nch, segs, st, csc_rank.get(), P.csr_idx.get(), CH, nullptr, 0, 0, samples.get());
    CoocArgs CA{P.rank_pair.get(), P.pair_start.get(), segs.ptr.get(), segs.seg.get(), segs.w.get(), P.csr_idx.get(), nullptr, 0, 0, Ic, CH, nch,
                0, Ic, 0, (int32_t)P.nnz, nullptr, 0, csr_pk.get(), nullptr, (uint32_t)std::min<int64_t>((int64_t)P.nnz * 4, 0xFFFFFFFFll)};
    CA.half = 1;
    CA.fx_scale = std::ldexp(1.0, fxk);
    CA.acc32 = acc32 ? 1 : 0;
    const int n_items = (int)cooc_item_count(Ic, CH, nch, true);
    DevBuf<int2> item_seg(ctx, (size_t)Ic * nch);
    DevBuf<int32_t> item_id(ctx, (size_t)Ic * nch), counter(ctx, 1);
    k_item_list<<<grid_for((int64_t)Ic * nch), 256, 0, st>>>(CA, item_seg.get(), item_id.get());
    FY_KERNEL_CHECK();
    CA.item_seg = item_seg.get();
    CA.item_id = item_id.get();
    counter.zero();
    CA.item_grab = cooc_item_grab(P.sum_deg2 / 2, n_items);
    MEpilogue ME{};
    ME.M = G;
    ME.ldm = ldm;
    ME.w2 = 1.0f;
    ME.fx_inv = std::ldexp(1.0, -fxk);
    t_tab.end(s0);
    const size_t s1 = t_walk.begin();
    launch_cooc_rm2(ctx, tune, true, CA, ME, n_items, counter.get(), st);
    t_walk.end(s1);
    sync(ctx);     // the tables of this scope go back to the allocator
    if (ms_tables) *ms_tables = t_tab.total_ms();
    if (ms_walk) *ms_walk = t_walk.total_ms();
    return true;
}

// one launch for the row kernels of several clusters (two-phase panel mode; fixed-point packed walk only): `L` = one CoocLaunch per
// cluster on the host, uploaded here; the launch shape (workgroup size, LDS) is that of the widest chunk among them
static void launch_cooc_rm2_multi(Context* ctx, const ScoreTune& tune, std::vector<CoocLaunch>& L, bool tail_role, DevBuf<CoocLaunch>& d_launch,
                                  DevBuf<int32_t>& d_counters, hipStream_t st, bool item_lists = false /* fill the item lists first */) {
    if (L.empty()) return;
    int max_chp = 0, max_items = 0, max_list = 0;
    for (size_t k = 0; k < L.size(); k++)      // (a null operand here is a fault on the GPU a moment later)
        if (!L[k].A.item_seg || !L[k].A.item_id || !L[k].E.M || !L[k].A.seg || !L[k].A.seg_ptr || (L[k].E.panel_cols && !L[k].E.Bmax64))
            FY_FAIL(FY_ERR_STATE, "internal: launch %zu of %zu of a batched row kernel has a null operand (item_seg %p item_id %p M %p seg %p ptr %p Bmax64 %p)", k, L.size(),
                    (const void*)L[k].A.item_seg, (const void*)L[k].A.item_id, (const void*)L[k].E.M, (const void*)L[k].A.seg, (const void*)L[k].A.seg_ptr, (const void*)L[k].E.Bmax64);
    for (auto& x : L) {
        x.A.acc_quarter = tune.cooc_planes ? cooc_lds_columns(x.A.CH) / 4 : 0;
        max_chp = std::max(max_chp, cooc_lds_columns(x.A.CH));
        max_items = std::max(max_items, x.n_items);
        max_list = std::max(max_list, x.A.nrows * x.A.nch);
    }
    const size_t lds = (size_t)max_chp * 8;
    const int by_lds = (int)std::max<size_t>(1, (160 * 1024 - 512) / (lds + 64));
    int block = tune.cooc_block;
    if (!block) block = by_lds >= 4 ? 256 : (by_lds >= 2 ? 512 : 1024);
    const int per_cu = std::max(1, std::min(by_lds, 2048 / block));
    const int gx = std::max(1, std::min(max_items, ctx->num_cus * per_cu));
    d_launch.alloc(ctx, L.size());
    d_counters.alloc(ctx, L.size());
    FY_HIP(hipMemcpyAsync(d_launch.get(), L.data(), L.size() * sizeof(CoocLaunch), hipMemcpyHostToDevice, st));
    FY_HIP(hipMemsetAsync(d_counters.get(), 0, L.size() * sizeof(int32_t), st));
    const dim3 grid((unsigned)gx, (unsigned)L.size());
    if (item_lists) {
        k_item_list_multi<<<dim3((unsigned)std::max(1, std::min(grid_for(max_list), 4096)), (unsigned)L.size()), 256, 0, st>>>(d_launch.get());
        FY_KERNEL_CHECK();
    }
    if (tail_role) k_cooc_rm2_multi<true, unsigned long long, 1><<<grid, block, lds, st>>>(d_launch.get(), d_counters.get());
    else k_cooc_rm2_multi<true, unsigned long long, 0><<<grid, block, lds, st>>>(d_launch.get(), d_counters.get());
    FY_KERNEL_CHECK();
}

// user slices (workgroups per column chunk) of a scoring launch over `nb` users and `chunks` column chunks: a wave walks up to
// users_per_wave users, but a small batch (one cluster of many: 3 250 users at 50 clusters) is cut finer so that the launch
// still has ~8 workgroups per CU -- with 16 users per wave such a launch had 102 workgroups for 256 CUs (1.4 ms per cluster
// for work that takes 0.14 ms of the one-cluster job)
static int score_slices(const Context* ctx, const ScoreTune& tune, int64_t nb, int chunks) {
    const int64_t coarse = ceil_div(nb, 4 * (int64_t)tune.users_per_wave), finest = ceil_div(nb, 4);
    const int64_t fill = ceil_div(8 * (int64_t)ctx->num_cus, std::max(1, chunks));
    return (int)std::max<int64_t>(1, std::min<int64_t>(tune.max_slices, std::min(finest, std::max(coarse, fill))));
}

// Super-blocks of the bound pass (k_score_sup): the fine 256-column blocks [seed_blocks, nblk) in at most 64 groups -- the first 48
// one block each (that is where survivors are: the columns are in popularity order and RM2 scores fall steeply with it), the rest
// in 16 groups of growing width.  first[s] .. first[s + 1] are the fine blocks of group s.
static void sup_block_map(int seed_blocks, int nblk, std::vector<int32_t>& first) {
    first.clear();
    const int R = std::max(0, nblk - seed_blocks);
    if (R <= 64) {
        for (int s = 0; s <= R; s++) first.push_back(seed_blocks + s);
        return;
    }
    for (int s = 0; s < 48; s++) first.push_back(seed_blocks + s);
    const int rem = R - 48;
    int64_t cum = 0;
    for (int k = 0; k < 16; k++) {            // widths ~ (k + 1): 1 + 2 + .. + 16 = 136 parts
        first.push_back(seed_blocks + 48 + (int32_t)((int64_t)rem * cum / 136));
        cum += k + 1;
    }
    first.push_back(nblk);
    for (size_t k = 1; k < first.size(); k++) first[k] = std::max(first[k], first[k - 1]);      // (monotone; a group may be empty)
}

// one cluster's launch plan
struct Plan {
    int c;
    int32_t Uc, sbase, pbase, Ic, a, b, CH, nch, q0, nq;
    int64_t ldm, B;
    bool pack24, prune, coop, half, panel;
    bool psym;       // symmetric panel mode: head rows x head columns by a half walk + mirror, head rows x tail columns not at all (k_panel_colmax)
    bool flat;       // one of many small unpruned clusters whose kernels run in ONE launch each (fy_rm2_kernels.hpp: FlatDesc)
    int32_t panel_cols, nsub;
    int64_t ldb64;
    // panel mode: the rows from p_eff on ("tail rows") are walked only over their first tail_chunks chunks (= the columns in
    // front of p_eff, a chunk boundary behind the panel); their block bounds behind p_eff come from k_tail_blocks' CSR
    int32_t tail_chunks, p_eff, tail_width;
    int32_t nblk;
    int64_t ldb;
};

#include "fy_rm2_coop.hpp"   // score_cluster_coop: a cluster scored by all ranks together (part of this translation unit)

static void validate_params(const fy_rm2_params* p) {
    if (!p) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "params is NULL");
    if (p->number_of_clusters <= 0) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "numberOfClusters must be > 0 (got %d)", p->number_of_clusters);
    if (p->number_of_items <= 0) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "numberOfItems must be > 0 (got %d)", p->number_of_items);
    if (!(p->lambda >= 0.0 && p->lambda <= 1.0)) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "lambda must be in [0, 1]");
    if (p->world <= 0 || p->rank < 0 || p->rank >= p->world) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "rank %d of world %d", p->rank, p->world);
    if (p->number_of_recommendations < 0) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "numberOfRecommendations must be >= 0");
}

// The row kernel's tables of ALL planned clusters in one pass over the CSR / CSC (no cooperative cluster among them): packed CSR,
// chunk offsets, the tail rows' block-compressed CSR, and both segment tables (main walk; tail-row bounds) of every cluster as
// ranges of ONE table -- one count launch, ONE scan, one host round trip for the size, one fill.  tc.segs / tc.segs_tail become views.
static void build_tables_all(Context* ctx, const Prepared& P, const std::vector<Plan>& plans, const std::vector<int32_t>& csr_range, bool use_pk,
                             TableCache& tc, hipStream_t st) {
    const size_t np = plans.size();
    std::vector<CoDesc> hco(np);
    std::vector<SegDesc> hsd;
    std::vector<int64_t> main_of(np, -1), tail_of(np, -1);
    int64_t co_total = 0, co_tail_total = 0, cnt_total = 0;
    int32_t max_slots = 1, max_f = 1, max_nq = 1;
    for (size_t pi = 0; pi < np; pi++) {
        const Plan& p = plans[pi];
        const bool tail = p.p_eff < p.Ic;
        hco[pi] = CoDesc{p.sbase, p.Uc, p.CH, p.nch, csr_range[2 * pi], csr_range[2 * pi + 1], p.p_eff, tail ? 1 : 0, co_total, co_tail_total};
        // panel mode: the tail rows are walked over their first tail_chunks chunks only; symmetric panel mode: so are the head rows,
        // and those are cut (Half::sym_rows / lim_from / lim_chunks) -- rows and chunks the item list never names get no segments
        SegDesc m{p.sbase, p.q0, p.nq, p.nch, p.CH, (p.half || p.psym) ? 1 : 0, 0, p.nch + 1, co_total, cnt_total, 0, p.psym ? p.p_eff : 0,
                  p.psym ? 0 : p.p_eff, (p.panel && p.tail_chunks > 0 && p.p_eff < p.Ic) ? p.tail_chunks : 0};
        main_of[pi] = (int64_t)hsd.size();
        hsd.push_back(m);
        cnt_total += (int64_t)p.nch * ((int64_t)p.nq + 1);
        co_total += (int64_t)p.Uc * (p.nch + 1);
        max_slots = std::max(max_slots, p.Uc);
        max_f = std::max(max_f, csr_range[2 * pi + 1] - csr_range[2 * pi]);
        max_nq = std::max(max_nq, p.nq);
    }
    for (size_t pi = 0; pi < np; pi++) {      // the tail tables behind the main ones (their chunk offsets are the k_tail_blocks ranges)
        const Plan& p = plans[pi];
        if (p.p_eff >= p.Ic) continue;
        SegDesc t{p.sbase, p.q0, p.nq, 1, 0, 0, p.p_eff, 2, hco[pi].co_tail_off, cnt_total, 1, 0, 0, 0};
        tail_of[pi] = (int64_t)hsd.size();
        hsd.push_back(t);
        cnt_total += (int64_t)p.nq + 1;
    }
    for (size_t pi = 0; pi < np; pi++)
        if (plans[pi].p_eff < plans[pi].Ic) { /* co_tail offsets were assigned above in plan order */ }
    // (co_tail_off: two entries per slot of every plan with tail rows, in plan order)
    {
        int64_t off = 0;
        for (size_t pi = 0; pi < np; pi++) {
            hco[pi].co_tail_off = off;
            if (plans[pi].p_eff < plans[pi].Ic) off += 2 * (int64_t)plans[pi].Uc;
        }
        co_tail_total = off;
        for (size_t pi = 0; pi < np; pi++)
            if (tail_of[pi] >= 0) hsd[(size_t)tail_of[pi]].co_off = hco[pi].co_tail_off;
    }
    if (cnt_total + 1 >= (int64_t)0x7FFFFFF0) FY_FAIL(FY_ERR_UNSUPPORTED, "segment tables of %zu clusters need %lld prefix entries (limit 2^31)", np, (long long)cnt_total);
    DevBuf<CoDesc> d_co(ctx, np);
    DevBuf<SegDesc> d_sd(ctx, hsd.size());
    FY_HIP(hipMemcpyAsync(d_co.get(), hco.data(), np * sizeof(CoDesc), hipMemcpyHostToDevice, st));
    FY_HIP(hipMemcpyAsync(d_sd.get(), hsd.data(), hsd.size() * sizeof(SegDesc), hipMemcpyHostToDevice, st));
    tc.g_co.alloc(ctx, (size_t)std::max<int64_t>(1, co_total));
    tc.g_co_tail.alloc(ctx, (size_t)std::max<int64_t>(1, co_tail_total));
    tc.g_cnt.alloc(ctx, (size_t)cnt_total + 1);
    tc.g_ptr.alloc(ctx, (size_t)cnt_total + 1);
    tc.g_start.alloc(ctx, (size_t)P.nnz + 1);
    const dim3 g_slots((unsigned)grid_for((int64_t)max_slots * 8, 256, 1024), (unsigned)np);
    if (use_pk) {
        k_pack_csr_multi<<<dim3((unsigned)grid_for(max_f, 256, 1024), (unsigned)np), 256, 0, st>>>(d_co.get(), P.csr_idx.get(), P.csr_r.get(), tc.csr_pk.get());
        FY_KERNEL_CHECK();
    }
    k_chunk_offsets_multi<<<g_slots, 256, 0, st>>>(d_co.get(), P.rowptr.get(), P.csr_idx.get(), tc.g_co.get());
    FY_KERNEL_CHECK();
    if (co_tail_total > 0) {
        k_tail_blocks_multi<<<dim3((unsigned)std::min<int>(max_slots, ctx->num_cus * 4), (unsigned)np), 256, 0, st>>>(d_co.get(), P.rowptr.get(), P.csr_idx.get(), P.csr_r.get(),
                                                                                                                  tc.y_pk.get(), tc.g_co_tail.get());
        FY_KERNEL_CHECK();
    }
    // main tables read g_co, tail tables g_co_tail: two launches of the count / fill kernels (the offsets are relative to either array)
    const unsigned n_main = (unsigned)np, n_tail = (unsigned)(hsd.size() - np);
    const dim3 g_q((unsigned)grid_for((int64_t)max_nq + 1, 256, 2048), n_main), g_qt((unsigned)grid_for((int64_t)max_nq + 1, 256, 2048), std::max(1u, n_tail));
    k_seg_counts_multi<<<g_q, 256, 0, st>>>(d_sd.get(), P.csc_slot.get(), tc.g_co.get(), tc.g_cnt.get(), tc.csc_rank.get(), P.csr_idx.get(), tc.g_start.get(), tc.samples.get());
    FY_KERNEL_CHECK();
    if (n_tail) {
        k_seg_counts_multi<<<g_qt, 256, 0, st>>>(d_sd.get() + np, P.csc_slot.get(), tc.g_co_tail.get(), tc.g_cnt.get(), tc.csc_rank.get(), P.csr_idx.get(), tc.g_start.get(), tc.samples.get());
        FY_KERNEL_CHECK();
    }
    FY_HIP(hipMemsetAsync(tc.g_cnt.get() + cnt_total, 0, sizeof(int32_t), st));
    exclusive_scan_i32(ctx, tc.g_cnt.get(), tc.g_ptr.get(), (size_t)cnt_total + 1, st);
    int32_t total = 0;
    FY_HIP(hipMemcpyAsync(&total, tc.g_ptr.get() + cnt_total, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    FY_HIP(hipStreamSynchronize(st));       // the ONE host round trip of the job's tables (d_co / d_sd's host copies are done too)
    tc.g_seg.alloc(ctx, (size_t)std::max(1, total));
    tc.g_w.alloc(ctx, (size_t)std::max(1, total));
    tc.total_segments = total;
    const float* csc_w = use_pk ? tc.csc_x_over_s.get() : tc.csc_x.get();
    const dim3 g_f((unsigned)grid_for((((int64_t)max_nq + 63) >> 6) * 64 * 4, 256, 4096), n_main), g_ft((unsigned)grid_for((((int64_t)max_nq + 63) >> 6) * 64, 256, 4096), std::max(1u, n_tail));
    k_seg_fill_multi<<<g_f, 256, 0, st>>>(d_sd.get(), P.csc_slot.get(), csc_w, tc.g_co.get(), tc.g_ptr.get(), tc.g_seg.get(), tc.g_w.get(), tc.csc_rank.get(),
                                          P.csr_idx.get(), tc.g_start.get());
    FY_KERNEL_CHECK();
    if (n_tail) {
        k_seg_fill_multi<<<g_ft, 256, 0, st>>>(d_sd.get() + np, P.csc_slot.get(), tc.csc_x_over_s.get(), tc.g_co_tail.get(), tc.g_ptr.get(), tc.g_seg.get(), tc.g_w.get(),
                                               tc.csc_rank.get(), P.csr_idx.get(), tc.g_start.get());
        FY_KERNEL_CHECK();
    }
    FY_HIP(hipStreamSynchronize(st));       // d_co / d_sd go back to the allocator
    for (size_t pi = 0; pi < np; pi++) {
        SegTable& m = tc.segs[pi];
        m.v_ptr = tc.g_ptr.get() + hsd[(size_t)main_of[pi]].cnt_off;
        m.v_seg = tc.g_seg.get();
        m.v_w = tc.g_w.get();
        if (tail_of[pi] >= 0) {
            SegTable& t = tc.segs_tail[pi];
            t.v_ptr = tc.g_ptr.get() + hsd[(size_t)tail_of[pi]].cnt_off;
            t.v_seg = tc.g_seg.get();
            t.v_w = tc.g_w.get();
        }
    }
}

fy_rm2_job* fy::rm2_prepare(Context* ctx, const fy_rm2_params* prm, const fy_ratings* R, int64_t n_map, const int32_t* map_user,
                            const int32_t* map_cluster, const int32_t* cluster_count) {
    validate_params(prm);
    if (n_map < 0 || (n_map > 0 && (!map_user || !map_cluster))) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "clustering map is NULL");
    EventTimer tm(ctx);
    const size_t span = tm.begin();
    const bool use_cache = !(prm->flags & FY_RM2_NO_CACHE);
    if (use_cache) {
        std::shared_ptr<RM2Static> have;
        {
            std::lock_guard<std::mutex> g(R->cache_mu);
            have = std::static_pointer_cast<RM2Static>(R->rm2_cache);
            // another clustering, rank or world: the old static set (CSR / CSC, statistics, every table of the row kernel) goes back to
            // the allocator BEFORE the new one is built -- peak HBM is one set plus the job's scratch, not two
            if (have && !have->matches(prm, n_map, map_user, map_cluster, cluster_count)) { R->rm2_cache.reset(); have.reset(); }
        }
        if (have) {      // warm: same ratings, same clustering, same share
            std::unique_ptr<fy_rm2_job> J(new fy_rm2_job(have));
            J->ctx = ctx;
            J->prm = *prm;
            J->from_cache = true;
            tm.end(span);
            sync(ctx);
            J->ms_prepare = tm.total_ms();
            return J.release();
        }
    }
    std::shared_ptr<RM2Static> fresh = std::make_shared<RM2Static>();
    fresh->K = prm->number_of_clusters;
    fresh->rank = prm->rank;
    fresh->world = prm->world;
    fresh->max_item = R->max_item;
    fresh->has_count = cluster_count != nullptr;
    if (n_map) { fresh->map_user.assign(map_user, map_user + n_map); fresh->map_cluster.assign(map_cluster, map_cluster + n_map); }
    if (cluster_count) fresh->cluster_count.assign(cluster_count, cluster_count + prm->number_of_clusters);
    std::unique_ptr<fy_rm2_job> J(new fy_rm2_job(fresh));
    J->ctx = ctx;
    J->prm = *prm;
    // Several ranks and at least as many clusters as ranks: this rank preps the ratings of its own clusters alone (the reference's
    // map-side partitioning by cluster, IntKeyPartitioner.java:15); the statistics are then exchanged by raw id.
    fy_ratings mine;
    DevBuf<int32_t> cl_table;
    SyncOnUnwind mine_guard(ctx->stream);       // (a failure below must not free `mine` under a queued kernel)
    RM2Static& S = *fresh;
    if (prm->world > 1 && ctx->tune.shard_prep)
        S.sharded = shard_ratings_by_cluster(ctx, R, prm->number_of_clusters, n_map, map_user, map_cluster, prm->rank, prm->world, mine, S.owner, cl_table);
    Prepared& P = J->P;
    if (S.sharded) {
        S.n_raw_users = (int64_t)R->max_user + 1;
        S.n_raw_items = (int64_t)R->max_item + 1;
        S.partial_raw.alloc(ctx, (size_t)S.exchange_len());
        S.partial_raw.zero();
        try {
            build_structure(ctx, &mine, prm->number_of_clusters, n_map, map_user, map_cluster, cluster_count, false, J->P, cl_table.get());
        } catch (const Failure& f) {
            if (f.code != FY_ERR_DUPLICATE_RATING && f.code != FY_ERR_CLUSTER_COUNT) throw;
            // only this rank's share shows it: the other ranks learn of it from the flag at the end of the exchange buffer
            // (fy_rm2_set_global_stats fails on EVERY rank then, none waits in a collective for a rank that has gone)
            S.deferred_code = f.code;
            S.deferred_msg = last_error();
            const double flag = (double)(-f.code);
            sync(ctx);
            FY_HIP(hipMemcpyAsync(S.partial_raw.get() + (S.exchange_len() - 1), &flag, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            tm.end(span);
            sync(ctx);
            J->ms_prepare = tm.total_ms();
            return J.release();
        }
        J->slot_lo = 0;
        J->slot_hi = P.nU;
    } else {
        build_structure(ctx, R, prm->number_of_clusters, n_map, map_user, map_cluster, cluster_count, false, J->P);
        rank_slot_range(P, prm->rank, prm->world, J->slot_lo, J->slot_hi);
    }
    J->partial.alloc(ctx, (size_t)P.nI + 1);
    J->partial.zero();
    // (sharded: the raw-id copy of the exchange buffer, queued behind the kernels that fill `partial` below)
    auto publish_raw = [&]() {
        if (!S.sharded) return;
        const int32_t n = std::max(P.nI + 1, P.nU);
        k_stats_to_raw<<<grid_for(n), 256, 0, ctx->stream>>>(P.nI, P.iid.get(), J->partial.get(), S.n_raw_items, P.nU, P.uid.get(), P.usum.get(),
                                                              S.partial_raw.get());
        FY_KERNEL_CHECK();
    };
    if (P.nnz > 0) {
        DevBuf<unsigned long long> counter(ctx, 1);
        counter.zero();
        J->usum_slot.alloc(ctx, (size_t)P.nU);
        J->deg_slot.alloc(ctx, (size_t)P.nU);
        DevBuf<double2> ud_slot(ctx, (size_t)P.nU);
        k_slot_user_arrays<<<grid_for(P.nU), 256, 0, ctx->stream>>>(P.nU, P.slot2du.get(), P.usum.get(), P.udeg.get(), J->usum_slot.get(),
                                                                    J->deg_slot.get(), ud_slot.get());
        FY_KERNEL_CHECK();
        // the walk's per-rating weights are written by the statistics pass (same gather): x / s_v for the packed walk, x for the plain one
        TableCache& tc0 = S.tables;
        const bool use_pk0 = ctx->tune.cooc_pk && P.ratings_fp16_exact;
        tc0.csc_x.alloc(ctx, (size_t)P.nnz);      // (x itself is also what the stray blocks of the panel mode are scored from)
        tc0.csc_x_over_s.alloc(ctx, use_pk0 ? (size_t)P.nnz : 1);
        tc0.have_x = true;
        J->b_rank.alloc(ctx, (size_t)P.nP);
        J->walk_rank.alloc(ctx, (size_t)P.nP);
        J->cnt_rank.alloc(ctx, (size_t)P.nP);
        J->fx_rank.alloc(ctx, 3 * (size_t)P.nP);
        // heavy columns (more than PAIR_HEAVY raters): at most nnz / PAIR_HEAVY of them, at most nnz / PAIR_HEAVY chunks
        const size_t max_heavy = (size_t)(P.nnz / PAIR_HEAVY) + 1;
        DevBuf<int32_t> heavy_col(ctx, max_heavy), heavy_base(ctx, max_heavy), chunk_col(ctx, max_heavy), heavy_counters(ctx, 2);
        DevBuf<PairAcc> chunk_acc(ctx, max_heavy);
        heavy_counters.zero();
        const PairPass PA{P.rank_pair.get(), P.pair_start.get(), P.pair_di.get(), P.csc_slot.get(), P.csc_r.get(), ud_slot.get(), J->slot_lo, J->slot_hi,
                          tc0.csc_x.get(), use_pk0 ? tc0.csc_x_over_s.get() : nullptr,
                          J->partial.get(), J->b_rank.get(), J->walk_rank.get(), J->cnt_rank.get(), J->fx_rank.get()};
        const PairHeavy PH{heavy_col.get(), heavy_base.get(), chunk_col.get(), heavy_counters.get(), chunk_acc.get()};
        if (P.nnz < 24 * (int64_t)P.nP) k_pair_pass<16><<<grid_for((int64_t)P.nP * 16, 256), 256, 0, ctx->stream>>>(P.nP, PA, PH);
        else k_pair_pass<64><<<grid_for((int64_t)P.nP * 64, 256), 256, 0, ctx->stream>>>(P.nP, PA, PH);
        FY_KERNEL_CHECK();
        k_pair_chunks<<<(unsigned)std::min<size_t>(max_heavy, (size_t)ctx->num_cus * 32), 256, 0, ctx->stream>>>(PA, PH);
        FY_KERNEL_CHECK();
        k_pair_finish<<<(unsigned)std::min<size_t>(ceil_div((int64_t)max_heavy, 64), 1024), 64, 0, ctx->stream>>>(PA, PH);
        FY_KERNEL_CHECK();
        if (J->slot_hi > J->slot_lo) {
            k_partial_total<<<grid_for(J->slot_hi - J->slot_lo), 256, 0, ctx->stream>>>(J->slot_lo, J->slot_hi, P.slot2du.get(),
                                                                                         P.usum.get(), counter.get());
            FY_KERNEL_CHECK();
        }
        k_store_total<<<1, 1, 0, ctx->stream>>>(counter.get(), J->partial.get() + P.nI);
        FY_KERNEL_CHECK();
        publish_raw();
        DevBuf<float> d_fx(ctx, 3 * (size_t)P.K);
        d_fx.zero();
        k_cluster_fx_bounds<<<dim3((unsigned)P.K, (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (int64_t)P.nP / ((int64_t)P.K * 1024)))), 256, 0, ctx->stream>>>(
            P.d_pcstart.get(), J->fx_rank.get(), d_fx.get());
        FY_KERNEL_CHECK();
        J->fx_bounds.resize(3 * (size_t)P.K);
        d2h(ctx, J->fx_bounds.data(), d_fx.get(), 3 * (size_t)P.K);
        tm.end(span);
        sync(ctx);
        J->ms_prepare = tm.total_ms();
        if (use_cache) { std::lock_guard<std::mutex> g(R->cache_mu); R->rm2_cache = fresh; }
        return J.release();
    }
    tm.end(span);
    sync(ctx);
    J->ms_prepare = tm.total_ms();
    if (use_cache) { std::lock_guard<std::mutex> g(R->cache_mu); R->rm2_cache = fresh; }
    return J.release();
}

void fy::rm2_set_global_stats(fy_rm2_job* J, const double* gathered, int32_t world) {
    Context* ctx = J->ctx;
    if (world != J->prm.world) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "gathered world %d != params.world %d", world, J->prm.world);
    const RM2Static& S = *J->S;
    const int64_t len = S.exchange_len();
    if (S.sharded) {
        if (world > 1024) FY_FAIL(FY_ERR_UNSUPPORTED, "more than 1024 ranks");
        // did any rank's share fail?  (the flags are read before anything is built from the sums)
        DevBuf<double> d_flags(ctx, (size_t)world);
        k_rank_flags<<<1, 1024, 0, ctx->stream>>>(world, len, gathered, d_flags.get());
        FY_KERNEL_CHECK();
        std::vector<double> flags((size_t)world);
        d2h(ctx, flags.data(), d_flags.get(), (size_t)world);
        sync(ctx);
        if (S.deferred_code) { set_error("%s", S.deferred_msg.c_str()); throw Failure{S.deferred_code}; }
        for (int r = 0; r < world; r++)
            if (flags[(size_t)r] != 0.0) {
                const int code = -(int)flags[(size_t)r];
                FY_FAIL(code, "rank %d found %s in the clusters it owns", r,
                        code == FY_ERR_DUPLICATE_RATING ? "two ratings with one (user, item) key" : "a clusteringCount that disagrees with the rated users");
            }
        J->stats_raw.alloc(ctx, (size_t)len);
        k_sum_gathered<<<grid_for(len), 256, 0, ctx->stream>>>(len, world, gathered, J->stats_raw.get());
        FY_KERNEL_CHECK();
        J->stats.alloc(ctx, (size_t)J->P.nI + 1);
        k_stats_from_raw<<<grid_for((int64_t)J->P.nI + 1), 256, 0, ctx->stream>>>(J->P.nI, J->P.iid.get(), J->stats_raw.get(), S.n_raw_items, J->stats.get());
        FY_KERNEL_CHECK();
        J->have_global = true;
        return;
    }
    J->stats.alloc(ctx, (size_t)len);
    k_sum_gathered<<<grid_for(len), 256, 0, ctx->stream>>>(len, world, gathered, J->stats.get());
    FY_KERNEL_CHECK();
    J->have_global = true;
}

fy_result* fy::rm2_score(fy_rm2_job* J) {
    Context* ctx = J->ctx;
    Prepared& P = J->P;
    const fy_rm2_params& prm = J->prm;
    hipStream_t st = ctx->stream;
    if (!J->have_global) {
        if (prm.world == 1) rm2_set_global_stats(J, J->partial.get(), 1);
        else if (J->have_coll) {   // the all-gather of the per-item statistics (jobs RM2-1 / RM2-2) through the installed collectives
            const int64_t len = J->S->exchange_len();
            J->gathered.alloc(ctx, (size_t)(len * prm.world));
            coll_all_gather(J, J->S->sharded ? J->S->partial_raw.get() : J->partial.get(), J->gathered.get(), len * (int64_t)sizeof(double), st);
            rm2_set_global_stats(J, J->gathered.get(), prm.world);
        } else
            FY_FAIL(FY_ERR_STATE, "world > 1: call fy_rm2_set_global_stats (or install fy_collectives) before fy_rm2_score");
    }
    std::unique_ptr<fy_result> R(new fy_result);
    R->ctx = ctx;
    R->kind = 0;
    R->st.nnz = P.nnz;
    R->st.n_users = P.nU;
    R->st.n_items = P.nI;
    R->st.ms_prepare = J->ms_prepare;
    R->st.prepared_from_cache = J->from_cache ? 1 : 0;
    EventTimer t_total(ctx), t_cooc(ctx), t_score(ctx), t_topn(ctx), t_tables(ctx), t_mirror(ctx);
    const size_t span_total = t_total.begin();
    size_t span_tables = t_tables.begin();
    if (P.nnz == 0) {
        t_total.end(span_total);
        sync(ctx);
        return R.release();
    }
    const int32_t nU = P.nU, nP = P.nP, nI = P.nI, K = P.K;
    const double lambda = prm.lambda;
    ScoreTune tune = score_tune(ctx);
    // tau_u is the N-th best of the seed scores: a seed of only a few N columns gives a weak threshold and many survivors
    // (Netflix shape, N = 100: 2.7 % of the blocks survive a 256-column seed, 0.1 % a 512-column one)
    // (round 4: also for long lists -- the reference's default is N = 1000, RMRecommenderDriver.java:95 -- whose seed is 5 N columns too:
    // 5120 of ML-25M's 59 047; up to round 3 the seed stopped at 1024 columns, whose 1000th best is no threshold at all, and such
    // jobs took the plain full pass)
    if (tune.seed_chunks == 0) tune.seed_chunks = (int)std::min<int64_t>(SEED_CHUNKS_MAX, std::max<int64_t>(1, ceil_div(5 * (int64_t)prm.number_of_recommendations, 256)));
    // lists the one-wave seed sort cannot hold (k_topn_seed: at most TOPN_SAMPLE seed columns, lists of at most TOPN_LONG items) take
    // k_topn_long in its seed / merge modes
    const bool long_seed = tune.seed_chunks * 256 > TOPN_SAMPLE || prm.number_of_recommendations > TOPN_LONG;
    const bool short_seed_ok = tune.seed_forced || 5 * (int64_t)prm.number_of_recommendations <= 4 * 256;     // (the cooperative path's limit)
    const bool pack24_allowed = tune.pack24 != 0;
    int64_t coop_pair_contribs = 0;   // cooperative clusters: ordered off-diagonal co-rating pairs of this rank's matrix rows
    bool any_coop = false;

    // ---- p(i|C), per-(cluster,item) statistics, per-rating values
    // (sharded prep: the job's own dense items are this rank's clusters' items; the result's itemColl is the global one, built at the end)
    DevBuf<double> icoll_mine;
    if (J->S->sharded) icoll_mine.alloc(ctx, nI); else R->d_icoll.alloc(ctx, nI);
    double* const d_icoll = J->S->sharded ? icoll_mine.get() : R->d_icoll.get();
    DevBuf<double> d_total(ctx, 1);
    k_item_coll<<<grid_for(nI), 256, 0, st>>>(nI, J->stats.get(), d_icoll, d_total.get());
    FY_KERNEL_CHECK();
    DevBuf<double> p_rank(ctx, nP);
    DevBuf<double>& b_rank = J->b_rank;
    DevBuf<float> a_rank(ctx, nP), b_rank32(ctx, nP);
    DevBuf<double2> pb_rank(ctx, nP);
    k_pair_p<<<grid_for(nP), 256, 0, st>>>(nP, P.rank_pair.get(), P.pair_di.get(), d_icoll, lambda, b_rank.get(), p_rank.get(), a_rank.get(),
                                           b_rank32.get(), pb_rank.get());
    FY_KERNEL_CHECK();
    DevBuf<float> csr_x(ctx, P.nnz), csr_e(ctx, P.nnz), csr_q(ctx, P.nnz);
    const bool use_pk = tune.cooc_pk && P.ratings_fp16_exact;   // packed CSR for the row kernel
    // the row kernel's tables live with the static part of the job (TableCache): a warm job finds them built
    TableCache& tc = J->S->tables;
    DevBuf<float>&csc_x = tc.csc_x, &csc_x_over_s = tc.csc_x_over_s;
    DevBuf<uint32_t>&csr_pk = tc.csr_pk, &y_pk = tc.y_pk;
    DevBuf<int32_t>& csc_rank = tc.csc_rank;
    std::vector<SegTable>&segs = tc.segs, &segs_tail = tc.segs_tail;
    // (normally written by fy_rm2_prepare's statistics pass; here only when the walk's kind changed between the two calls)
    if (!tc.have_x || csc_x.size() != (size_t)P.nnz || csc_x_over_s.size() != (use_pk ? (size_t)P.nnz : 1)) {
        tc.valid = tc.have_x = false;
        csc_x.alloc(ctx, P.nnz);
        csc_x_over_s.alloc(ctx, use_pk ? (size_t)P.nnz : 1);
        k_csc_x<<<grid_for(P.nnz), 256, 0, st>>>(P.nnz, P.csc_slot.get(), P.csc_r.get(), J->usum_slot.get(), csc_x.get(),
                                                 use_pk ? csc_x_over_s.get() : nullptr);
        FY_KERNEL_CHECK();
        tc.have_x = true;
    }
    // packed (24-bit) matrix rows only where bandwidth matters: small clusters keep exact fp32 rows (their scores are small, and the
    // reference's own fixture is asserted with an ABSOLUTE 1e-4, T/util/HadoopIntegrationTest.java:53).  A packed cluster's matrix is
    // scaled by 2^-c so that every entry is < 1 (FY_P24_SHIFT): G[j][i] = w2 sum_v (r_vj / s_v^2) r_vi <= w2 * (largest column sum
    // of r / s^2) * (largest rating), the bounds of the fixed-point scale.
    std::vector<float> h_gscale((size_t)K, 1.0f);
    std::vector<int32_t> h_cshift((size_t)K, 0);
    std::vector<char> cluster_pack24((size_t)K, 0);
    for (int c = 0; c < K; c++) {
        const int32_t Ic_c = P.pcstart[c + 1] - P.pcstart[c];
        if (!pack24_allowed || Ic_c < tune.pack24_min_items || J->fx_bounds.size() < 3 * (size_t)(c + 1)) continue;
        const double gmax = (1.0 - lambda) * (1.0 - lambda) * (double)J->fx_bounds[3 * (size_t)c] * (double)J->fx_bounds[3 * (size_t)c + 2];
        if (!(gmax >= 0.0) || !std::isfinite(gmax)) continue;                     // unusable bound: fp32 rows
        const int cs = gmax > 0.0 ? std::max(0, std::ilogb(gmax) + 1) : 0;
        if (cs > 64) continue;
        cluster_pack24[c] = 1;
        h_cshift[c] = cs;
        h_gscale[c] = std::ldexp(1.0f, -cs);
    }
    DevBuf<float> d_gscale(ctx, (size_t)K);
    DevBuf<int32_t> d_cshift(ctx, (size_t)K);
    h2d(ctx, d_gscale.get(), h_gscale.data(), (size_t)K);
    h2d(ctx, d_cshift.get(), h_cshift.data(), (size_t)K);
    // The per-rating values (x, e, q: what the SCORING kernels read): with the packed walk nothing in front of the scoring reads them, and
    // what runs until then -- the table kernels, then the one-cluster job's row kernel with ONE workgroup per CU beside its 157 KB of LDS
    // accumulators -- leaves wave slots and most of the memory system idle.  They are computed on a side stream; the plain one-cluster
    // flow joins behind its row kernel, every other flow (several lanes, cooperative ranks, flat batches) before its lanes start.
    struct SideValues {
        Context* ctx;
        hipStream_t sv = nullptr;
        hipEvent_t in = nullptr, out = nullptr;
        bool launched = false, joined = true;
        ~SideValues() {     // (a failed job: the side kernel must not outlive the arrays it writes)
            if (!joined && sv) (void)hipStreamSynchronize(sv);
            if (in) (void)hipEventDestroy(in);
            if (out) (void)hipEventDestroy(out);
        }
    } side{ctx};
    auto launch_values = [&](hipStream_t main_stream, bool beside) {
        if (side.launched) return;
        side.launched = true;
        hipStream_t vs = main_stream;
        if (beside && tune.overlap_values) {
            if (ctx->aux.empty()) {
                hipStream_t x;
                FY_HIP(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
                ctx->aux.push_back(x);
            }
            side.sv = ctx->aux[0];
            FY_HIP(hipEventCreateWithFlags(&side.in, hipEventDisableTiming));
            FY_HIP(hipEventCreateWithFlags(&side.out, hipEventDisableTiming));
            FY_HIP(hipEventRecord(side.in, main_stream));
            FY_HIP(hipStreamWaitEvent(side.sv, side.in, 0));
            vs = side.sv;
        }
        k_csr_values<<<grid_for((int64_t)nU * 64, 256), 256, 0, vs>>>(nU, P.rowptr.get(), P.csr_idx.get(), P.csr_r.get(), P.slot2du.get(),
                                                                       P.ucluster.get(), P.usum.get(), P.d_csize.get(), P.d_pcstart.get(),
                                                                       pb_rank.get(), lambda, d_gscale.get(), csr_x.get(), csr_e.get(), csr_q.get());
        FY_KERNEL_CHECK();
        if (vs != main_stream) {
            FY_HIP(hipEventRecord(side.out, side.sv));
            side.joined = false;
        }
    };
    auto join_values = [&](hipStream_t s) {
        if (side.joined) return;
        side.joined = true;
        FY_HIP(hipStreamWaitEvent(s, side.out, 0));
    };
    // (the packed walk reads the packed CSR; the PLAIN walk's row kernel reads x itself: no overlap there.  Measured on one box, ML-25M
    // shape, ms per cold job: side stream from here 20.08, launched right in front of the row kernel 20.15, main stream 20.30 / 20.48 --
    // the table kernels in between wait on the L2's request rate and on dependent loads, and leave more room than the row kernel.)
    launch_values(st, use_pk);

    t_tables.end(span_tables);
    // ---- which users this rank emits lists for
    J->count_balanced = false;
    if (prm.world > 1 && !J->S->sharded && J->have_coll && tune.coop && tune.prune && pack24_allowed) {     // (a cooperative cluster must be a packed one: checked per plan below)
        int nonempty = 0, c1 = -1;
        for (int c = 0; c < K; c++)
            if (P.csize[c] > 0) { nonempty++; c1 = c; }
        if (nonempty == 1) {   // one neighbourhood: it is scored cooperatively when it is big enough for the branch and bound
            const int32_t Ic1 = P.pcstart[c1 + 1] - P.pcstart[c1];
            J->count_balanced = Ic1 >= tune.pack24_min_items && Ic1 >= tune.prune_min_items && ceil_div(Ic1, PRUNE_BLOCK) < 0xFFFF &&
                                P.nU >= prm.world && short_seed_ok && !long_seed;
        }
    }
    int32_t own_lo, own_hi;
    owner_range(J, prm.rank, own_lo, own_hi);
    // ---- per-user meta for this rank's slots, output offsets
    const int32_t lo = own_lo, hi = own_hi, nmine = hi - lo;
    DevBuf<double> pvpi(ctx, (size_t)nmine + 1);
    DevBuf<int32_t> n_out(ctx, (size_t)nmine + 1), out_off(ctx, (size_t)nmine + 1);
    DevBuf<unsigned long long> counters(ctx, 2);
    counters.zero();
    n_out.zero();
    if (nmine > 0) {
        k_user_meta<<<grid_for(nmine), 256, 0, st>>>(lo, hi, P.slot2du.get(), P.uid.get(), P.ucluster.get(), P.udeg.get(),
                                                      P.d_csize.get(), P.d_pcstart.get(), prm.number_of_items,
                                                      prm.number_of_recommendations, prm.filter_users, d_cshift.get(), pvpi.get(), n_out.get(), counters.get());
        FY_KERNEL_CHECK();
    }
    exclusive_scan_i32(ctx, n_out.get(), out_off.get(), (size_t)nmine + 1);
    const int64_t n_recs = fetch(ctx, out_off.get() + nmine);
    {
        unsigned long long hc[2];
        d2h(ctx, hc, counters.get(), 2);
        sync(ctx);
        R->st.log_terms = (int64_t)hc[0];
        R->st.users_scored = (int64_t)hc[1];
    }
    R->n = n_recs;
    R->st.recs = n_recs;
    R->d_key0.alloc(ctx, (size_t)n_recs);
    R->d_key1.alloc(ctx, (size_t)n_recs);
    R->d_value.alloc(ctx, (size_t)n_recs);
    R->d_aux.alloc(ctx, (size_t)n_recs);

    // ---- clusters that hold users of this rank
    int64_t max_Ic = 0;
    for (int c = 0; c < K; c++) {
        if (P.csize[c] == 0) continue;
        R->st.n_clusters_nonempty++;
        const int32_t a = std::max(lo, P.ucstart[c]), b = std::min(hi, P.ucstart[c + 1]);
        if (a < b) max_Ic = std::max<int64_t>(max_Ic, P.pcstart[c + 1] - P.pcstart[c]);
    }
    const int32_t eff_top = (int32_t)std::min<int64_t>(prm.number_of_recommendations, max_Ic);
    if (eff_top > TOPN_MAX)
        FY_FAIL(FY_ERR_UNSUPPORTED, "min(numberOfRecommendations, items per cluster) = %d exceeds the top-N kernel limit %d", eff_top, TOPN_MAX);
    if (n_recs > 0 && max_Ic > 0) {
        const int64_t ws = prm.workspace_bytes > 0 ? prm.workspace_bytes : tune.workspace_default;
        const int max_ch_lds = tune.cooc_max_ch;   // fp64 accumulators in LDS
        cooc_rm2_allow_lds();
        stray_allow_lds();

        // ---- per-cluster plan; the clusters are spread over up to four "lanes" (HIP streams with their own M / score
        // scratch): the tail of one cluster's launches -- its heaviest user sits on a single wave for milliseconds, and
        // most of its M rows have a handful of raters -- overlaps the next clusters' work instead of idling the chip.
        std::vector<Plan> plans;
        constexpr int VEC = 4;   // floats per lane of the scoring kernel: column chunk = 256 items
        for (int c = 0; c < K; c++) {
            Plan p{};
            p.c = c;
            p.Uc = P.csize[c];
            if (p.Uc == 0) continue;
            p.sbase = P.ucstart[c];
            p.pbase = P.pcstart[c];
            p.Ic = P.pcstart[c + 1] - p.pbase;
            p.a = std::max(lo, p.sbase);
            p.b = std::min(hi, p.sbase + p.Uc);
            if (p.a >= p.b || p.Ic == 0) continue;
            p.ldm = round_up(p.Ic, 256);
            p.pack24 = cluster_pack24[c] != 0;
            // (the item ids of the row kernel hold the chunk in 8 bits, k_item_list: at most 255 chunks per row -- a forced
            // FY_COOC_MAX_CH too small for that is widened)
            pick_chunks(p.Ic, std::max<int>(max_ch_lds, (int)round_up(ceil_div(p.Ic, 255), 256)), p.CH, p.nch);
            if (p.nch >= 256) FY_FAIL(FY_ERR_UNSUPPORTED, "cluster %d: %d items need %d column chunks (limit 255)", c, p.Ic, p.nch);
            p.q0 = P.cluster_q[c];
            p.nq = P.cluster_q[c + 1] - p.q0;
            // branch-and-bound over 256-column blocks: only where the matrix is big enough for the bound pass to pay
            p.nblk = (int32_t)ceil_div(p.Ic, PRUNE_BLOCK);
            p.ldb = round_up(p.nblk, 256);
            // (the threshold is the N-th best of at most 1024 seed scores: for lists longer than ~200 items it is too weak --
            // N = 1000 at ML-25M shape: 77 % of the blocks survive and the three passes cost twice the plain one)
            // (and clusters of a few hundred users are not pruned at all: their seed thresholds are weak -- at 400 clusters of ML-25M
            // shape, 406 users each, 16-44 % of the blocks survive and the plain full pass is 1.6x faster than any pruned variant)
            // (long lists: the seed is 5 N columns; where that is more than a third of the cluster's items the bound has nothing left to
            // exclude and the plain full pass is taken)
            p.prune = tune.prune && p.pack24 && p.Ic >= tune.prune_min_items && p.nblk < 0xFFFF && p.Uc >= tune.prune_min_users &&
                      (tune.seed_forced || 3 * (int64_t)tune.seed_chunks * 256 <= (int64_t)p.Ic);
            if (J->count_balanced && !p.prune) FY_FAIL(FY_ERR_STATE, "internal: count-balanced ownership without a cooperative cluster");
            // all ranks hold users of this cluster and can talk to each other: score it together, every rank with its
            // share of the matrix rows (score_cluster_coop)
            p.coop = false;
            if (p.prune && tune.coop && short_seed_ok && !long_seed && ((prm.world > 1 && J->have_coll) || tune.coop_force)) {
                p.coop = true;
                for (int k = 0; k < prm.world; k++) {
                    int32_t lo_k, hi_k;
                    owner_range(J, k, lo_k, hi_k);
                    if (std::max(lo_k, p.sbase) >= std::min(hi_k, p.sbase + p.Uc)) p.coop = false;
                }
            }
            // symmetric walk: packed rows only (small clusters keep exact fp32 rows and the plain walk); a cooperative rank
            // owns whole rows of the matrix, so it walks them whole
            p.half = tune.cooc_half && p.pack24 && !p.coop && p.nch < 256;
            p.panel = false;
            p.panel_cols = 0;
            p.tail_chunks = 0;
            p.p_eff = p.Ic;
            p.tail_width = 0;
            p.nsub = (int32_t)ceil_div(p.Ic, 64);
            p.ldb64 = round_up(p.nsub, 256);
            plans.push_back(p);
        }
        {   // column-panel mode: many pruned clusters on this rank (the reference's regime: numberOfClusters ~ 50)
            int n_pruned = 0;
            for (auto& p : plans) n_pruned += (p.prune && !p.coop) ? 1 : 0;
            // Long lists are pruned only where a few big clusters keep their dense matrices.  Measured at 50 clusters of ML-25M shape with
            // N = 1000 (round 4): 41 % of the (user, block) pairs survive the bound of a 3 250-user cluster and 10 M of them lie behind
            // the panel -- 14.9 s per job against 0.67 s for the plain full pass; one cluster: 14 % survive, 271 against 471 ms.
            // (and on dense per-cluster matrices instead of panels: 29 568 184 of 29 568 241 blocks survive -- a 3 250-user neighbourhood's
            // 1000th best score is no threshold -- every cluster falls back to the full pass, 957 ms)
            if (long_seed && n_pruned >= tune.panel_min_clusters && !tune.seed_forced) {
                for (auto& p : plans)
                    if (!p.coop) p.prune = false;
                n_pruned = 0;
            }
            // A cluster that takes the plain full pass reads its WHOLE matrix.  The symmetric walk + mirror pass pays where the walk is
            // bound by its pair visits (one cluster of ML-25M shape: 6.5e9 visits, 8 ms; the mirror 2.6 ms); a cluster of many (50
            // clusters: 1.3e8 visits each for 2.6e9 matrix elements) is bound by the rows it WRITES -- 1.65 ms for the half matrix plus 4.3 ms
            // to mirror 7.8 GB, against ~3.3 ms for the full walk: such clusters walk full rows and skip the mirror.
            for (auto& p : plans) {
                const double deg2 = (size_t)p.c < P.cluster_deg2.size() ? (double)P.cluster_deg2[p.c] : (double)P.sum_deg2;
                if (!p.prune && !p.coop && p.half && tune.full_walk_sparse && deg2 < (double)p.Ic * (double)p.Ic) p.half = false;
            }
            if (n_pruned >= tune.panel_min_clusters)
                for (auto& p : plans)
                    if (p.prune && !p.coop && use_pk && p.nch < 256 && p.nsub < 0xFFFF && p.Uc <= STRAY_UCAP) {
                        p.panel = true;
                        p.half = false;      // a row's block maxima need the whole row
                        // (the item ids of the row kernel hold the chunk in 8 bits)
                        pick_chunks(p.Ic, std::min<int>(max_ch_lds, std::max<int>(tune.panel_max_ch, (int)round_up(ceil_div(p.Ic, 255), 256))), p.CH, p.nch);
                        // smaller clusters keep a wider panel: their survivors reach further down the popularity order (measured,
                        // ML-25M shape, ms per job with 4096 / 8192 columns: 50 clusters of 3250 users 124 / 141, 100 clusters of
                        // 1625 users 295 / 227, 200 clusters of 812 users 919 / 509 -- the difference is blocks behind the panel)
                        const int64_t want_cols = (int64_t)tune.panel_cols * (p.Uc < tune.panel_wide_below_users ? 2 : 1);
                        p.panel_cols = (int32_t)std::min<int64_t>(p.ldm, std::max<int64_t>(round_up(want_cols, 256), (int64_t)tune.seed_chunks * 256));
                        // (symmetric panel mode needs the head rows = the panel's columns: whole chunks of a width that divides the panel)
                        // Either the chunk width is re-picked so that it divides the panel, or -- where that would take too many rows out of
                        // the head (the rows with EXACT sub-block maxima; a tail row's bounds behind the panel are sums, and looser: Netflix
                        // shape in 50 clusters, chunks of 3584 columns: 7168 head rows; with 4096 head rows 3867 instead of 58 blocks survive
                        // behind the panel and the job takes 713 instead of 175 ms) -- the panel is widened to the chunks that cover it.
                        // (Widening from 1.25 x instead of 1.5 x the panel: 200 clusters of ML-25M shape 285 -> 299 ms, 25 clusters 59.8 -> 62.6 ms.)
                        const int64_t p_eff_chunks = std::min<int64_t>(ceil_div(p.panel_cols, p.CH) * (int64_t)p.CH, p.Ic);
                        if (tune.panel_sym && p.panel_cols % p.CH != 0 && 2 * p_eff_chunks > 3 * (int64_t)p.panel_cols && p_eff_chunks % 256 == 0 && p_eff_chunks < p.Ic)
                            p.panel_cols = (int32_t)p_eff_chunks;
                        if (tune.panel_sym && p.panel_cols % p.CH != 0) {
                            const int32_t lim = std::min<int>(max_ch_lds, std::max<int>(tune.panel_max_ch, (int)round_up(ceil_div(p.Ic, 255), 256)));
                            for (int32_t parts = 1; parts <= 8; parts++) {
                                const int32_t w = p.panel_cols / parts;
                                if (p.panel_cols % parts == 0 && w % 256 == 0 && w <= lim && ceil_div(p.Ic, w) < 256) { p.CH = w; p.nch = (int32_t)ceil_div(p.Ic, w); break; }
                            }
                        }
                        p.tail_chunks = (int32_t)ceil_div(p.panel_cols, p.CH);
                        p.p_eff = (int32_t)std::min<int64_t>((int64_t)p.tail_chunks * p.CH, p.Ic);
                        if (p.p_eff % 256 != 0 || p.tail_chunks >= p.nch) { p.p_eff = p.Ic; p.tail_chunks = 0; }   // one chunk, or a ragged one: no tail rows
                        p.tail_width = (int32_t)(p.ldb64 - p.p_eff / 64);
                    }
        }
        const int64_t flat_budget = tune.flat_budget > 0 ? tune.flat_budget : (int64_t)std::min<uint64_t>(ctx->total_mem / 4, (uint64_t)std::max<int64_t>(ws, (int64_t)8 << 30));
        auto flat_need = [](const Plan& p) {
            return (int64_t)p.Ic * p.ldm * (p.pack24 ? 3 : 4) + (int64_t)(p.b - p.a) * p.ldm * 4 + (int64_t)p.Ic * p.nch * 12 + 4096;
        };
        {   // flat batch: the small unpruned clusters of a multi-cluster job, every kernel of their chain ONE launch for all of them
            // (FlatDesc, fy_rm2_kernels.hpp).  The matrices and score rows of the clusters of one batch are resident together; a job
            // whose clusters do not fit the budget takes several batches.
            int n_flat = 0;
            const bool can = tune.flat_batch && plans.size() > 1 && use_pk && tune.cooc_fx && !tune.cooc_f32 && !J->fx_bounds.empty() &&
                             prm.number_of_recommendations <= TOPN_LONG;
            for (auto& p : plans) {
                p.flat = false;
                if (!can || p.prune || p.coop || p.panel) continue;
                if (fx_exponent(&J->fx_bounds[3 * (size_t)p.c]) < 0) continue;
                if (flat_need(p) > flat_budget) continue;
                p.flat = true;
                n_flat++;
            }
            if (n_flat < 2)
                for (auto& p : plans) p.flat = false;
            for (auto& p : plans) p.psym = false;
            for (auto& p : plans)
                if (p.flat) p.half = false;      // (a mirror pass per cluster would be two more launches each)
        }
        bool any_panel = false;
        for (auto& p : plans) any_panel = any_panel || p.panel;
        // Two phases for panel-mode jobs (round 3): the matrix panels of ALL clusters are built first, back to back on the main stream
        // (persistent row kernels that fill every CU's LDS gain nothing from running beside another cluster's), each into its own
        // buffers; then the clusters' light scoring kernels overlap on the lanes.  With one set of buffers per LANE (round 2) two lanes
        // were the optimum and mostly waited for each other's row kernels.  Needs every cluster's panel resident: 45 GB at 50 clusters
        // of ML-25M shape.
        int64_t panel_bytes = 0;
        int n_panel = 0;
        for (auto& p : plans)
            if (p.panel) { n_panel++; panel_bytes += (int64_t)p.Ic * p.panel_cols * 3 + (int64_t)p.Ic * p.ldb64 * 7 + 64; }
        // (more panels than fit a third of the HBM: the clusters are taken in GROUPS, each through all phases -- 100 clusters of ML-25M
        // shape keep 127 GB of panels)
        const bool two_phase = tune.panel_two_phase && n_panel >= 2;
        std::vector<int> group_of(plans.size(), 0);
        int n_groups = 1;
        if (two_phase) {
            const int64_t limit = tune.panel_group_bytes > 0 ? tune.panel_group_bytes : (int64_t)(ctx->total_mem / 3);
            int64_t in_group = 0;
            int g = 0;
            for (size_t pi = 0; pi < plans.size(); pi++) {
                const Plan& p = plans[pi];
                if (!p.panel) continue;
                const int64_t need = (int64_t)p.Ic * p.panel_cols * 3 + (int64_t)p.Ic * p.ldb64 * 7 + (int64_t)(p.b - p.a) * p.ldb64 * 7 + (int64_t)p.Ic * p.nch * 12;
                if (in_group > 0 && in_group + need > limit) { g++; in_group = 0; }
                in_group += need;
                group_of[pi] = g;
            }
            n_groups = g + 1;
        }
        (void)panel_bytes;
        // Symmetric panel mode (two-phase jobs, batched fixed-point row kernels): G is symmetric, so (1) inside the panel's square
        // [0, p_eff)^2 the head rows are walked like the one-cluster job's -- only the columns behind the row, k_mirror_* fills the
        // rest -- and (2) the head rows are not walked over the tail columns at all: those co-ratings are the tail rows' with the head
        // columns, which the tail rows walk and STORE (Gp[j][i], j >= p_eff > i), and the only thing the head rows needed them for,
        // the maxima of their 64-column sub-blocks, are column maxima of the stored panel (k_panel_colmax).  Half the pair visits.
        if (two_phase && tune.panel_sym && tune.panel_multi_launch && use_pk && tune.cooc_fx && !tune.cooc_f32 && !J->fx_bounds.empty())
            for (auto& p : plans)
                p.psym = p.panel && p.tail_chunks > 0 && p.p_eff < p.Ic && p.p_eff == p.panel_cols && p.p_eff % 256 == 0 && p.a == p.sbase && p.b == p.sbase + p.Uc &&
                         fx_exponent(&J->fx_bounds[3 * (size_t)p.c]) >= 0;
        // (lanes: one-phase panel mode 2 -- more lanes only queue behind each other's row kernels; two-phase 8 -- only light kernels are left
        // on the lanes: measured at 50 clusters, ms per job: 2 lanes 98.0, 4: 95.9, 8: 93.1)
        const int want_lanes = tune.lanes_forced ? tune.lanes : (two_phase ? std::max(tune.lanes, 8) : (any_panel ? std::min(tune.lanes, tune.panel_lanes) : tune.lanes));
        const int NS = (int)std::min<size_t>(plans.size() > 1 ? (size_t)want_lanes : 1, plans.size());
        // (the side stream's per-rating values, see above: only the plain one-cluster flow lets them run beside its row kernel)
        if (!(NS == 1 && plans.size() == 1 && !plans[0].coop && !plans[0].flat && !plans[0].panel)) join_values(st);
        struct Lane {
            hipStream_t st;
            DevBuf<float> M, S;
            DevBuf<int32_t> overflow, any_overflow;
            // branch and bound
            DevBuf<float> Bmax, amax, bmax, UB, tau;
            DevBuf<float> Gp, Bmax64, amax64, bmax64;    // column-panel mode
            DevBuf<uint32_t> Brep;
            DevBuf<uint16_t> surv;
            DevBuf<uint8_t> surv_mask;     // panel mode: which 64-column sub-blocks of a surviving block passed the bound
            DevBuf<int2> strayT;           // co-rater tables of k_score_stray
            DevBuf<int2> stray_items;
            DevBuf<int32_t> n_heavy;       // k_count_heavy
            DevBuf<int32_t> need;          // lazy mirror: column blocks with survivors
            DevBuf<float> Bsup, asup, bsupb, UBs;      // super-block bounds (k_score_sup)
            DevBuf<double> head32, tail32;             // unrounded fp64 rows / columns the refinement pass reads (k_refine_rows)
            DevBuf<int32_t> colmap;
            DevBuf<int32_t> sup_first;
            DevBuf<int32_t> n_quads, quad_prefix;
            DevBuf<char> scan_tmp;         // temporary storage of the lane's scans
            DevBuf<int2> item_seg, item_seg_t;      // (_t: the tail-row bound launch of a cluster whose row kernels are batched)
            DevBuf<int32_t> item_id, item_id_t;
            DevBuf<float> Ssurv;   // packed scores of the surviving blocks (pruned clusters)
        };
        std::vector<Lane> lanes((size_t)NS);
        {
            size_t is_el = 1;
            for (auto& p : plans)
                if (!p.flat) is_el = std::max(is_el, (size_t)p.Ic * p.nch);
            size_t m_el = 1, s_el = 1, ov_el = 1, bm_el = 1, ub_el = 1, am_el = 1, gp_el = 1, b64_el = 1, a64_el = 1;
            for (auto& p : plans) {
                p.B = std::min<int64_t>(std::max<int64_t>(1, (ws / NS) / (p.ldm * 4)), p.b - p.a);
                // pruned clusters keep only the seed columns of a score row (the survivors' scores are packed, see below):
                // all users of the rank in one batch
                const int64_t seed_cols = (int64_t)std::min<int64_t>(ceil_div(p.Ic, 256), tune.seed_chunks) * 256;
                if (p.prune || p.flat) p.B = p.b - p.a;
                if (p.coop || p.flat) continue;   // allocate for themselves
                if (p.panel) {
                    gp_el = std::max(gp_el, (size_t)p.Ic * p.panel_cols * 3 / 4 + 4);
                    b64_el = std::max(b64_el, (size_t)p.Ic * p.ldb64 * 3 / 4 + 4);
                    a64_el = std::max(a64_el, (size_t)p.ldb64);

                    s_el = std::max(s_el, (size_t)(p.B * seed_cols));
                    ov_el = std::max(ov_el, (size_t)p.B);
                    ub_el = std::max(ub_el, (size_t)(p.B * p.ldb64));
                    continue;
                }
                m_el = std::max(m_el, (size_t)(p.Ic * p.ldm));
                s_el = std::max(s_el, (size_t)(p.B * (p.prune ? seed_cols : p.ldm)));
                ov_el = std::max(ov_el, (size_t)p.B);
                if (p.prune) {
                    bm_el = std::max(bm_el, (size_t)(p.Ic * p.ldb));
                    ub_el = std::max(ub_el, (size_t)(p.B * p.ldb));
                    am_el = std::max(am_el, (size_t)p.ldb);
                }
            }
            if (NS > 1 && ctx->aux.size() < (size_t)NS) {
                while (ctx->aux.size() < (size_t)NS) {
                    hipStream_t x;
                    FY_HIP(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
                    ctx->aux.push_back(x);
                }
            }
            for (int l = 0; l < NS; l++) {
                Lane& L = lanes[l];
                L.st = NS > 1 ? ctx->aux[l] : st;
                L.M.alloc(ctx, m_el);
                L.S.alloc(ctx, s_el);
                L.overflow.alloc(ctx, ov_el);
                L.any_overflow.alloc(ctx, 1);
                L.n_heavy.alloc(ctx, 1);
                L.Bmax.alloc(ctx, bm_el);
                L.amax.alloc(ctx, am_el);
                L.bmax.alloc(ctx, am_el);
                L.Gp.alloc(ctx, two_phase ? 1 : gp_el);
                L.Bmax64.alloc(ctx, two_phase ? 1 : b64_el);
                L.Brep.alloc(ctx, gp_el > 1 && !two_phase ? b64_el * 4 / 3 + 4 : 1);     // (b64_el counts floats for 3-byte entries)
                L.amax64.alloc(ctx, two_phase ? 1 : a64_el);
                L.bmax64.alloc(ctx, two_phase ? 1 : a64_el);
                L.UB.alloc(ctx, ub_el);
                L.tau.alloc(ctx, ov_el);
                L.surv.alloc(ctx, ub_el);
                L.surv_mask.alloc(ctx, gp_el > 1 ? ub_el : 1);
                L.n_quads.alloc(ctx, ov_el + 1);
                L.quad_prefix.alloc(ctx, ov_el + 1);
                L.item_seg.alloc(ctx, is_el);
                L.item_id.alloc(ctx, is_el);
            }
        }
        DevBuf<unsigned long long> prune_counters(ctx, 6);   // ... [5] list rows scored again by the refinement pass   // [0] surviving blocks, [1] log terms evaluated by the three pruned passes, [2] users sent to k_topn_select, [3] stray blocks (panel mode)
        prune_counters.zero();
        int64_t prune_blocks_total = 0, prune_seed_terms_cols = 0, coop_survived = 0, fallback_survived = 0;
        for (auto& p : plans) any_coop = any_coop || p.coop;
        // segment tables of the row kernel, one per cluster (built on the main stream before the lanes fork, or by the lanes): kept
        // with the job's static part -- a job over the same ratings, clustering and launch plan re-uses them
        bool any_tail = false, any_half = false;
        size_t co_all = 1;
        for (auto& p : plans) {
            any_tail = any_tail || p.p_eff < p.Ic;
            co_all = std::max(co_all, (size_t)p.Uc * (p.nch + 1));
            any_half = any_half || p.half || p.p_eff < p.Ic;
        }
        std::vector<int32_t> sig;
        sig.push_back(use_pk ? 1 : 0);
        sig.push_back(plans.size() > 1 && !any_coop ? 1 : 0);
        for (auto& p : plans) {
            const int32_t v[8] = {p.c, p.CH, p.nch, p.half ? 1 : 0, p.panel ? 1 : 0, p.p_eff, p.tail_chunks, (p.coop ? 1 : 0) | (p.flat ? 2 : 0) | (p.psym ? 4 : 0)};
            sig.insert(sig.end(), v, v + 8);
        }
        const bool tables_cached = tc.valid && tc.sig == sig;
        R->st.tables_from_cache = tables_cached ? 1 : 0;
        span_tables = t_tables.begin();
        if (!tables_cached) {
            tc.valid = false;
            segs.clear();
            segs_tail.clear();
            segs.resize(plans.size());
            segs_tail.resize(plans.size());
            csr_pk.alloc(ctx, use_pk ? (size_t)P.nnz : 1);
            y_pk.alloc(ctx, any_tail ? (size_t)P.nnz : 1);     // k_tail_blocks
            csc_rank.alloc(ctx, any_half ? (size_t)P.nnz : 1);     // row (rank inside its cluster) of every CSC entry
            if (any_half) {
                k_csc_rank<<<grid_for(P.nnz), 256, 0, st>>>(P.nnz, P.csc_pair.get(), P.pair_rank.get(), csc_rank.get());
                FY_KERNEL_CHECK();
                build_csr_samples(ctx, P.nnz, P.csr_idx.get(), tc.samples, st);
            }
        }
        // first / last CSR entry of every planned cluster (slots are cluster-major): one round trip for all of them
        std::vector<int32_t> csr_range(2 * plans.size() + 2, 0);
        {
            std::vector<int32_t> where(2 * plans.size());
            for (size_t pi = 0; pi < plans.size(); pi++) {
                where[2 * pi] = plans[pi].sbase;
                where[2 * pi + 1] = plans[pi].sbase + plans[pi].Uc;
            }
            gather_to_host_i32(ctx, P.rowptr.get(), where, csr_range.data());
        }
        // One cluster's tables: packed CSR with chunk-relative indices for its CH, chunk offsets, segment table (+ the tail rows'
        // one-chunk table over the block-compressed CSR in panel mode).  `co` = scratch for p.Uc * (p.nch + 1) offsets.
        const bool bounded_tables = false;     // (round 2: tables sized by upper bounds inside the lanes; the lanes build no tables any more)
        auto build_tables = [&](size_t pi, hipStream_t ts, int32_t* co) {
            if (tables_cached) return;
            const Plan& p = plans[pi];
            const int32_t f0 = csr_range[2 * pi], f1 = csr_range[2 * pi + 1];
            if (use_pk && f1 > f0) {
                k_pack_csr<<<grid_for(f1 - f0), 256, 0, ts>>>(f0, f1, p.CH, P.csr_idx.get(), P.csr_r.get(), csr_pk.get());
                FY_KERNEL_CHECK();
            }
            if (p.coop) return;
            build_chunk_offsets(ctx, P.rowptr.get(), P.csr_idx.get(), p.sbase, p.Uc, p.CH, p.nch, co, ts);
            // In a lane the tables are sized by an upper bound instead of a count read back from the device (an entry of rater v
            // has at most n_v / 64 + nch segments): the host does not wait, the lane's queue stays full
            const int64_t deg2_c = (size_t)p.c < P.cluster_deg2.size() ? P.cluster_deg2[p.c] : 0;
            const int64_t seg_bound = bounded_tables && deg2_c > 0 ? deg2_c / 64 + (int64_t)p.nq * p.nch + 64 : 0;
            build_segments(ctx, P.csc_slot.get(), use_pk ? csc_x_over_s.get() : csc_x.get(), co, p.sbase, p.q0, p.nq, p.nch, segs[pi], ts,
                           p.half ? csc_rank.get() : nullptr, P.csr_idx.get(), p.CH, nullptr, 0, seg_bound, p.half ? tc.samples.get() : nullptr);
            if (p.p_eff < p.Ic) {
                k_tail_blocks<<<std::min<int>(p.Uc, ctx->num_cus * 16), 256, 0, ts>>>(p.sbase, p.Uc, p.p_eff, P.rowptr.get(), P.csr_idx.get(), P.csr_r.get(),
                                                                          y_pk.get(), co);
                FY_KERNEL_CHECK();
                build_segments(ctx, P.csc_slot.get(), csc_x_over_s.get(), co, p.sbase, p.q0, p.nq, 1, segs_tail[pi], ts, nullptr, nullptr, 0,
                               csc_rank.get(), p.p_eff, bounded_tables && deg2_c > 0 ? deg2_c / 64 + (int64_t)p.nq + 64 : 0);
            }
        };
        // With several lanes and no cooperative cluster the tables are built by the lane that uses them, right before the row
        // kernel: table building is mostly host round trips (sizes of the segment tables), 23 ms for 50 clusters during which the
        // chip idled; in a lane they hide behind the other lanes' kernels.  (A cooperative job packs every cluster's CSR up front.)
        // Several clusters and none of them cooperative: ONE table over all of them, built here on the main stream (build_tables_all).
        // (Round 2 let every lane build its cluster's tables right before the row kernel, to hide their host round trips behind the
        // other lanes: 20 ms of tables at 50 clusters.)
        const bool all_at_once = plans.size() > 1 && !any_coop;
        const bool lazy_tables = false;
        std::vector<DevBuf<int32_t>> co_lane((size_t)NS);
        if (!all_at_once)
            for (auto& b : co_lane) b.alloc(ctx, co_all);
        if (all_at_once) {
            if (!tables_cached) build_tables_all(ctx, P, plans, csr_range, use_pk, tc, st);
        } else if (!lazy_tables)
            for (size_t pi = 0; pi < plans.size(); pi++) build_tables(pi, st, co_lane[0].get());
        t_tables.end(span_tables);
        // ---- flat batch (Plan::flat): matrix build, scores and lists of all those clusters, one launch per kernel, on the main stream
        DevBuf<char> flatM;
        DevBuf<float> flatS;
        DevBuf<int2> flat_seg;
        DevBuf<int32_t> flat_id, flat_ov, flat_flags, flat_cnt;
        DevBuf<CoocLaunch> d_flat_launch;
        DevBuf<FlatDesc> d_flat[2];
        for (size_t first = 0; first < plans.size();) {
            std::vector<CoocLaunch> fb;
            std::vector<FlatDesc> fd[2];       // [0] fp32 rows, [1] 24-bit rows
            SyncOnUnwind fb_fd_guard(st);      // (their uploads are queued below; the batch's own synchronisation is at its end)
            size_t m_bytes = 0, s_el = 0, i_el = 0, u_el = 0, n_flat = 0, last = first;
            int64_t batch_bytes = 0;
            for (; last < plans.size(); last++) {
                const Plan& p = plans[last];
                if (!p.flat) continue;
                if (n_flat && batch_bytes + flat_need(p) > flat_budget) break;
                batch_bytes += flat_need(p);
                m_bytes += round_up((int64_t)p.Ic * p.ldm * (p.pack24 ? 3 : 4), 256);
                s_el += (size_t)(p.b - p.a) * p.ldm;
                i_el += (size_t)p.Ic * p.nch;
                u_el += (size_t)(p.b - p.a);
                n_flat++;
            }
            if (n_flat) {
                flatM.alloc(ctx, m_bytes);
                flatS.alloc(ctx, s_el);
                flat_seg.alloc(ctx, i_el);
                flat_id.alloc(ctx, i_el);
                flat_ov.alloc(ctx, u_el);
                flat_flags.alloc(ctx, 2 * n_flat);       // per cluster: n_heavy, any_overflow
                size_t m_at = 0, s_at = 0, i_at = 0, u_at = 0, k = 0;
                int max_grid[2] = {0, 0}, max_users[2] = {0, 0};
                for (size_t pi = first; pi < last; pi++) {
                    const Plan& p = plans[pi];
                    if (!p.flat) continue;
                    const int c = p.c;
                    const int32_t nb = p.b - p.a;
                    float* const Mc = reinterpret_cast<float*>(flatM.get() + m_at);
                    float* const Sc = flatS.get() + s_at;
                    CoocArgs CA{P.rank_pair.get(), P.pair_start.get(), segs[pi].ptr_(), segs[pi].seg_(), segs[pi].w_(), P.csr_idx.get(),
                                csr_x.get(), p.pbase, p.sbase, p.Ic, p.CH, p.nch, 0, p.Ic, p.q0, p.nq, nullptr, 0, csr_pk.get(), nullptr,
                                (uint32_t)std::min<int64_t>((int64_t)P.nnz * 4, 0xFFFFFFFFll)};
                    const int fxk = fx_exponent(&J->fx_bounds[3 * (size_t)c]);
                    CA.fx_scale = std::ldexp(1.0, fxk);
                    const double w2s = (1.0 - lambda) * (1.0 - lambda) * (double)h_gscale[(size_t)c];
                    MEpilogue ME{Mc, p.ldm, (float)w2s, std::ldexp(w2s, -fxk), p.pack24 ? 1 : 0, nullptr, p.ldb, 0, 0, nullptr, p.ldb64, nullptr};
                    const int n_items = (int)cooc_item_count(p.Ic, p.CH, p.nch, false);
                    CA.item_seg = flat_seg.get() + i_at;
                    CA.item_id = flat_id.get() + i_at;
                    CA.item_grab = cooc_item_grab((size_t)c < P.cluster_deg2.size() ? P.cluster_deg2[c] : P.sum_deg2, n_items);
                    fb.push_back(CoocLaunch{CA, ME, n_items, 0});

                    const int n_chunks = (int)ceil_div(p.Ic, 64 * VEC);
                    // (the chip is filled by all clusters of the batch together: ~8 work-groups per CU over the whole launch)
                    const int64_t fill = ceil_div(8 * (int64_t)ctx->num_cus, (int64_t)std::max(1, n_chunks) * (int64_t)n_flat);
                    const int n_slices = (int)std::max<int64_t>(1, std::min<int64_t>(tune.max_slices, std::min<int64_t>(ceil_div(nb, 4), std::max<int64_t>(ceil_div(nb, 4 * (int64_t)tune.users_per_wave), fill))));
                    FlatDesc d{};
                    ScoreArgs& SA = d.SA;
                    SA.M = Mc; SA.ldm = p.ldm; SA.Ic = p.Ic; SA.a_rank = a_rank.get() + p.pbase; SA.b_rank = b_rank32.get() + p.pbase;
                    SA.rb_off = P.rowptr.get() + p.sbase;
                    SA.csr_idx = P.csr_idx.get(); SA.csr_e = csr_e.get(); SA.csr_q = csr_q.get();
                    SA.pvpi = pvpi.get(); SA.n_out = n_out.get(); SA.slot_lo = lo; SA.slot_base = p.sbase; SA.slot0 = p.a; SA.n_users = nb;
                    SA.S = Sc; SA.ldS = p.ldm; SA.n_slices = n_slices; SA.n_chunks = n_chunks;
                    d.n_heavy = flat_flags.get() + 2 * k;
                    d.any_overflow = flat_flags.get() + 2 * k + 1;
                    SA.n_heavy = tune.score_heavy > 0 ? d.n_heavy : nullptr;
                    d.heavy_thresh = tune.score_heavy > 0 ? std::min(tune.score_heavy, 32) : 0x7FFFFFFF;
                    d.TA = TopNArgs{Sc, p.ldm, p.Ic, n_out.get(), out_off.get(), P.rank_item_raw.get() + p.pbase, P.slot2du.get(), P.uid.get(), lo, p.a, c,
                                    R->d_key0.get(), R->d_key1.get(), R->d_value.get(), R->d_aux.get(), 0, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr};
                    d.overflow = flat_ov.get() + u_at;
                    d.score_grid = n_chunks * n_slices;
                    d.n_users = nb;
                    const int f = p.pack24 ? 1 : 0;
                    max_grid[f] = std::max(max_grid[f], d.score_grid);
                    max_users[f] = std::max(max_users[f], nb);
                    fd[f].push_back(d);
                    m_at += (size_t)round_up((int64_t)p.Ic * p.ldm * (p.pack24 ? 3 : 4), 256);
                    s_at += (size_t)nb * p.ldm;
                    i_at += (size_t)p.Ic * p.nch;
                    u_at += (size_t)nb;
                    k++;
                }
                const size_t sp = t_cooc.begin(st);
                launch_cooc_rm2_multi(ctx, tune, fb, false, d_flat_launch, flat_cnt, st, true);
                t_cooc.end(sp, st);
                R->st.cooc_launches++;
                for (int f = 0; f < 2; f++) {
                    if (fd[f].empty()) continue;
                    const unsigned ny = (unsigned)fd[f].size();
                    d_flat[f].alloc(ctx, fd[f].size());
                    FY_HIP(hipMemcpyAsync(d_flat[f].get(), fd[f].data(), fd[f].size() * sizeof(FlatDesc), hipMemcpyHostToDevice, st));
                    const size_t ss = t_score.begin(st);
                    k_count_heavy_multi<<<(ny + 63) / 64, 64, 0, st>>>(d_flat[f].get(), (int32_t)ny);
                    FY_KERNEL_CHECK();
                    if (f) k_score_multi<4, true, 8><<<dim3((unsigned)max_grid[f], ny), 256, 0, st>>>(d_flat[f].get(), P.csr_idx.get(), csr_e.get(), csr_q.get(), pvpi.get(), n_out.get());
                    else k_score_multi<4, false, 8><<<dim3((unsigned)max_grid[f], ny), 256, 0, st>>>(d_flat[f].get(), P.csr_idx.get(), csr_e.get(), csr_q.get(), pvpi.get(), n_out.get());
                    FY_KERNEL_CHECK();
                    t_score.end(ss, st);
                    R->st.score_launches++;
                    const size_t tt = t_topn.begin(st);
                    k_topn_fast_multi<<<dim3((unsigned)max_users[f], ny), 256, 0, st>>>(d_flat[f].get(), tune.force_select);
                    FY_KERNEL_CHECK();
                    k_topn_select_multi<<<dim3((unsigned)max_users[f], ny), 256, 0, st>>>(d_flat[f].get(), prune_counters.get() + 2);
                    FY_KERNEL_CHECK();
                    t_topn.end(tt, st);
                }
                FY_HIP(hipStreamSynchronize(st));     // the host vectors behind the descriptor uploads leave scope here
            }
            first = last;
        }
        struct PanelBuf {
            DevBuf<float> Gp, Bmax64, amax64, bmax64;
            DevBuf<uint32_t> Brep;
            DevBuf<double> head32, tail32;     // (refinement pass: per cluster, like the panels -- built in phase 1, read in phase 3)
        };
        struct PanelPtrs {
            float *Gp, *Bmax64;
            uint32_t* Brep;
            float *amax64, *bmax64;
        };
        std::vector<PanelBuf> pbuf(two_phase ? plans.size() : 0);
        std::vector<Lane> plane(two_phase ? plans.size() : 0);
        auto group_buffers = [&](int grp) {
            for (size_t pi = 0; pi < plans.size(); pi++) {
                const Plan& p = plans[pi];
                if (!p.panel || group_of[pi] != grp) continue;
                pbuf[pi].Gp.alloc(ctx, (size_t)p.Ic * p.panel_cols * 3 / 4 + 4);
                pbuf[pi].Bmax64.alloc(ctx, (size_t)p.Ic * p.ldb64 * 3 / 4 + 4);
                pbuf[pi].Brep.alloc(ctx, (size_t)p.Ic * p.ldb64 + 4);
                pbuf[pi].amax64.alloc(ctx, (size_t)p.ldb64);
                pbuf[pi].bmax64.alloc(ctx, (size_t)p.ldb64);
            }
        // ... and its own scoring scratch, so that NO host round trip separates the clusters: phase 2 queues seed + bound pass, select, second
        // bound and the survivor count of every cluster, ONE wait reads all counts (pinned host memory), phase 3 queues the survivor
        // passes and top-N.  (With one scratch set per lane the host waited for every cluster's count before it could queue the next
        // cluster of that lane: 50 round trips during which the other lanes ran dry.)
            for (size_t pi = 0; pi < plans.size(); pi++) {
                const Plan& p = plans[pi];
                if (!p.panel || group_of[pi] != grp) continue;
                Lane& W = plane[pi];
                const size_t nbp = (size_t)(p.b - p.a);
                const size_t seed_cols = (size_t)std::min<int64_t>(ceil_div(p.Ic, 256), tune.seed_chunks) * 256;
                W.st = lanes[pi % NS].st;
                W.M.alloc(ctx, 1); W.Bmax.alloc(ctx, 1); W.amax.alloc(ctx, 1); W.bmax.alloc(ctx, 1);
                W.Gp.alloc(ctx, 1); W.Bmax64.alloc(ctx, 1); W.Brep.alloc(ctx, 1); W.amax64.alloc(ctx, 1); W.bmax64.alloc(ctx, 1);
                W.item_seg.alloc(ctx, (size_t)p.Ic * p.nch); W.item_id.alloc(ctx, (size_t)p.Ic * p.nch);
                W.item_seg_t.alloc(ctx, (size_t)std::max(1, p.Ic - p.p_eff)); W.item_id_t.alloc(ctx, (size_t)std::max(1, p.Ic - p.p_eff));
                W.S.alloc(ctx, nbp * seed_cols);
                W.overflow.alloc(ctx, nbp);
                W.any_overflow.alloc(ctx, 1);
                W.n_heavy.alloc(ctx, 1);
                W.UB.alloc(ctx, nbp * (size_t)p.ldb64);
                W.tau.alloc(ctx, nbp);
                W.surv.alloc(ctx, nbp * (size_t)p.ldb64);
                W.surv_mask.alloc(ctx, nbp * (size_t)p.ldb64);
                W.n_quads.alloc(ctx, nbp + 1);
                W.quad_prefix.alloc(ctx, nbp + 1);
            }
        };
        // Error path: anything thrown below (an allocation, a launch, a collective) unwinds the lanes' buffers, the segment
        // tables and the per-job arrays back into the caching allocator while kernels of OTHER lanes may still be reading
        // them.  The guard drains every lane and the main stream first (members are destroyed in reverse order of
        // declaration: `lanes`, `segs` and the DevBufs above were declared before it, so it runs before they are released).
        // (The flat batch above runs on the main stream alone, in front of the guard: its uploads' sources have their own SyncOnUnwind.)
        // (host-side sources of uploads and the device buffers of the batched launches: declared in FRONT of the guard, so that the
        // guard -- which drains every lane and the main stream -- is destroyed before them on every path)
        std::vector<CoocLaunch> batch_main, batch_tail;      // phase 1: the row kernels of all panel clusters, launched together
        DevBuf<CoocLaunch> d_batch_main, d_batch_tail;
        DevBuf<int32_t> d_cnt_main, d_cnt_tail;
        DevBuf<PanelDesc> d_panel_desc;
        std::vector<PanelDesc> hpd;                          // (lives as long as its upload may be in flight)
        struct LaneGuard {
            Context* ctx;
            hipEvent_t fork = nullptr;
            int32_t* pinned = nullptr;
            ~LaneGuard() {
                for (hipStream_t x : ctx->aux) (void)hipStreamSynchronize(x);
                (void)hipStreamSynchronize(ctx->stream);
                if (fork) (void)hipEventDestroy(fork);
                if (pinned) (void)hipHostFree(pinned);
            }
        } guard{ctx};
        if (two_phase) FY_HIP(hipHostMalloc(reinterpret_cast<void**>(&guard.pinned), plans.size() * sizeof(int32_t), hipHostMallocDefault));
        // phase 0: every cluster start to end on its lane.  two_phase: phase 1 = the panels of all panel-mode clusters on the main stream,
        // phase 2 = everything else on the lanes.
        for (int grp = 0; grp < n_groups; grp++) {
        if (two_phase) {
            if (grp > 0) {       // the previous group's kernels have drained (below): its panels and scratch go back to the allocator
                for (size_t pi = 0; pi < plans.size(); pi++)
                    if (plans[pi].panel && group_of[pi] == grp - 1) { pbuf[pi] = PanelBuf(); plane[pi] = Lane(); }
            }
            group_buffers(grp);
            batch_main.clear();
            batch_tail.clear();
            hpd.clear();
        }
        for (int phase = two_phase ? 1 : 0; phase <= (two_phase ? 3 : 0); phase++) {
        const auto host_t0 = std::chrono::steady_clock::now();
        struct PhaseClock {
            const std::chrono::steady_clock::time_point t0;
            int grp, phase;
            bool on;
            ~PhaseClock() {
                if (on) fprintf(stderr, "[fy] group %d phase %d: host queued for %.3f ms\n", grp, phase, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
        } phase_clock{host_t0, grp, phase, tune.debug_sync != 0};
        if (phase == 3) {             // every cluster's survivor count has been queued: one wait for all of them
            for (int l = 0; l < NS; l++) FY_HIP(hipStreamSynchronize(lanes[l].st));
            FY_HIP(hipStreamSynchronize(st));
        }
        if ((phase == 0 || phase == 2) && NS > 1) {   // the lanes start after everything queued on the main stream so far
            if (guard.fork) { FY_HIP(hipEventDestroy(guard.fork)); guard.fork = nullptr; }
            FY_HIP(hipEventCreateWithFlags(&guard.fork, hipEventDisableTiming));
            FY_HIP(hipEventRecord(guard.fork, st));
            for (int l = 0; l < NS; l++) FY_HIP(hipStreamWaitEvent(lanes[l].st, guard.fork, 0));
        }

        for (size_t pi = 0; pi < plans.size(); pi++) {
            const Plan& p = plans[pi];
            if (p.flat) continue;                  // done above
            if (group_of[pi] != grp) continue;     // (clusters outside panel mode are in group 0)
            if ((phase == 1 || phase == 3) && !p.panel) continue;
            Lane& L = phase == 1 ? lanes[0] : (two_phase && p.panel ? plane[pi] : lanes[pi % NS]);
            hipStream_t ls = phase == 1 ? st : L.st;
            const bool do_build = !(phase >= 2 && p.panel), do_score = phase != 1;
            const PanelPtrs PP = two_phase && p.panel ? PanelPtrs{pbuf[pi].Gp.get(), pbuf[pi].Bmax64.get(), pbuf[pi].Brep.get(), pbuf[pi].amax64.get(), pbuf[pi].bmax64.get()}
                                                      : PanelPtrs{L.Gp.get(), L.Bmax64.get(), L.Brep.get(), L.amax64.get(), L.bmax64.get()};
            auto checkpoint = [&](const char* what) {
                if (tune.debug_sync != 1) return;
                const hipError_t e = hipDeviceSynchronize();
                fprintf(stderr, "[fy] group %d/%d phase %d plan %zu (cluster %d): %s -> %s\n", grp, n_groups, phase, pi, p.c, what, hipGetErrorString(e));
                fflush(stderr);
            };
            checkpoint("start");
            if (p.panel && (!PP.Gp || !PP.Bmax64 || !PP.Brep || !PP.amax64 || !PP.bmax64 || (two_phase && (!L.S.get() || !L.UB.get() || !L.n_quads.get()) && phase != 1)))
                FY_FAIL(FY_ERR_STATE, "internal: cluster %d (plan %zu, group %d of %d, phase %d) has no panel buffers", p.c, pi, grp, n_groups, phase);
            const int c = p.c;
            const int32_t sbase = p.sbase, pbase = p.pbase, Ic = p.Ic, a = p.a, b = p.b, CH = p.CH, nch = p.nch;
            const int64_t ldm = p.ldm;
            const bool pack24 = p.pack24;
            if (p.coop) {
                CoopShared X{J, R.get(), &tune, b_rank32.get(), a_rank.get(), use_pk ? csc_x_over_s.get() : csc_x.get(), csr_x.get(), csr_e.get(), csr_q.get(),
                             use_pk ? csr_pk.get() : nullptr,
                             n_out.get(), out_off.get(), pvpi.get(), lo, &t_cooc, &t_score, &t_topn, prune_counters.get(),
                             &prune_blocks_total, &prune_seed_terms_cols, &coop_survived, &coop_pair_contribs, d_cshift.get(), h_gscale[(size_t)c]};
                score_cluster_coop(X, p, ls);
                continue;
            }

            if (lazy_tables) {
                const size_t stb = t_tables.begin(ls);
                build_tables(pi, ls, co_lane[pi % NS].get());
                t_tables.end(stb, ls);
            }
            // -- M build
            CoocArgs CA{P.rank_pair.get(), P.pair_start.get(), segs[pi].ptr_(), segs[pi].seg_(), segs[pi].w_(), P.csr_idx.get(),
                        csr_x.get(), pbase, sbase, Ic, CH, nch, 0, Ic, p.q0, p.nq, nullptr, 0, use_pk ? csr_pk.get() : nullptr, nullptr,
                        (uint32_t)std::min<int64_t>((int64_t)P.nnz * 4, 0xFFFFFFFFll)};
            const int fxk = (use_pk && tune.cooc_fx && !J->fx_bounds.empty()) ? fx_exponent(&J->fx_bounds[3 * (size_t)c]) : -1;
            CA.fx_scale = fxk >= 0 ? std::ldexp(1.0, fxk) : 0.0;
            const double w2s = (1.0 - lambda) * (1.0 - lambda) * (double)h_gscale[(size_t)c];     // (1-l)^2 and the packed format's 2^-c
            MEpilogue ME{L.M.get(), ldm, (float)w2s, fxk >= 0 ? std::ldexp(w2s, -fxk) : 0.0,
                         pack24 ? 1 : 0, (p.prune && !p.panel) ? L.Bmax.get() : nullptr, p.ldb, 0,
                         p.panel ? p.panel_cols : 0, p.panel ? PP.Bmax64 : nullptr, p.ldb64, p.panel ? PP.Brep : nullptr};
            if (p.panel) ME.M = PP.Gp;
            // refinement (k_refine_rows): packed clusters keep the unrounded fp32 values of their first 256 rows (and, in symmetric panel
            // mode, of the first 256 columns of the tail rows)
            const bool refine = tune.refine && pack24 && !p.coop && fxk >= 0 && J->S->max_item >= 0 && J->S->max_item < (1 << 28) && Ic >= 8;
            // (as many head rows / columns as the seed is wide, 256 .. 1024: N = 50 -> 256, N = 100 -> 512; a multiple of 4)
            const int32_t head_rows = (int32_t)std::min<int64_t>(Ic & ~3, std::max<int64_t>(256, std::min<int64_t>(1024, (int64_t)tune.seed_chunks * 256)));
            const int64_t ld_head = p.psym ? (int64_t)p.p_eff : ldm;
            DevBuf<double>& H32 = (two_phase && p.panel) ? pbuf[pi].head32 : L.head32;
            DevBuf<double>& T32 = (two_phase && p.panel) ? pbuf[pi].tail32 : L.tail32;
            if (refine && do_build) {
                H32.alloc(ctx, (size_t)head_rows * ld_head + 4);
                if (p.psym) T32.alloc(ctx, (size_t)std::max(1, Ic - p.p_eff) * head_rows + 4);
                ME.head32 = H32.get();
                ME.ld_head = ld_head;
                ME.head_rows = head_rows;
                ME.tail32 = p.psym ? T32.get() : nullptr;
                ME.tail_from = p.p_eff;
            }
            if (do_build) {
            if (p.panel) {
                R->st.panel_clusters++;
                FY_HIP(hipMemsetAsync(PP.Bmax64, 0, (size_t)Ic * p.ldb64 * 3, ls));
                k_block_amax<<<grid_for(p.ldb64), 256, 0, ls>>>(Ic, (int32_t)p.ldb64, a_rank.get() + pbase, b_rank32.get() + pbase, PP.amax64,
                                                               PP.bmax64, 64);
                FY_KERNEL_CHECK();
            } else if (p.prune) {
                FY_HIP(hipMemsetAsync(L.Bmax.get(), 0, (size_t)Ic * p.ldb * 3, ls));
                k_block_amax<<<grid_for(p.ldb), 256, 0, ls>>>(Ic, (int32_t)p.ldb, a_rank.get() + pbase, b_rank32.get() + pbase, L.amax.get(), L.bmax.get());
                FY_KERNEL_CHECK();
            }
            const size_t sp = t_cooc.begin(ls);
            const bool batched = phase == 1 && tune.panel_multi_launch && use_pk && fxk >= 0 && !tune.cooc_f32;     // (k_cooc_rm2_multi)
            if (p.p_eff < Ic) {   // panel mode: bounds of the tail rows behind p_eff (one item per row, see k_tail_blocks)
                CoocArgs CB{P.rank_pair.get(), P.pair_start.get(), segs_tail[pi].ptr_(), segs_tail[pi].seg_(), segs_tail[pi].w_(), P.csr_idx.get(),
                            csr_x.get(), pbase, sbase, Ic, p.tail_width, 1, p.p_eff, Ic - p.p_eff, p.q0, p.nq, nullptr, 0, y_pk.get(), nullptr, CA.pk_bytes};
                CB.fx_scale = CA.fx_scale;
                MEpilogue MB{reinterpret_cast<float*>(reinterpret_cast<char*>(PP.Bmax64) + (size_t)(p.p_eff / 64) * 3), p.tail_width, ME.w2, ME.fx_inv, 1,
                             nullptr, 0, 0, 0, nullptr, 0, nullptr, p.ldb64, 1};
                int2* const tseg = batched ? plane[pi].item_seg_t.get() : L.item_seg.get();
                int32_t* const tid = batched ? plane[pi].item_id_t.get() : L.item_id.get();
                if (!batched) {      // (batched: k_item_list_multi, in front of the launch)
                    k_item_list<<<grid_for((int64_t)(Ic - p.p_eff)), 256, 0, ls>>>(CB, tseg, tid);
                    FY_KERNEL_CHECK();
                }
                CB.item_seg = tseg;
                CB.item_id = tid;
                CB.item_grab = 8;      // a tail row's bound item is a handful of segments
                if (batched) batch_tail.push_back(CoocLaunch{CB, MB, Ic - p.p_eff, 0});
                else {
                    FY_HIP(hipMemsetAsync(L.any_overflow.get(), 0, sizeof(int32_t), ls));
                    launch_cooc_rm2(ctx, tune, use_pk, CB, MB, Ic - p.p_eff, L.any_overflow.get(), ls);
                    R->st.cooc_launches++;
                }
                CA.tail_row0 = p.p_eff;
                CA.tail_chunks = p.tail_chunks;
            }
            {
                CA.half = (p.half || p.psym) ? 1 : 0;
                if (p.psym) { CA.half_rows = p.p_eff; CA.head_chunks = p.tail_chunks; }
                const int n_items = p.psym ? (int)(cooc_half_item_index(p.p_eff - 1, p.tail_chunks - 1, CH, p.tail_chunks) + 1 + (int64_t)(Ic - p.p_eff) * p.tail_chunks)
                                    : CA.tail_chunks > 0 ? (int)((int64_t)p.p_eff * nch + (int64_t)(Ic - p.p_eff) * p.tail_chunks)
                                                         : (int)cooc_item_count(Ic, CH, nch, p.half);
                int2* const mseg = batched ? plane[pi].item_seg.get() : L.item_seg.get();
                int32_t* const mid = batched ? plane[pi].item_id.get() : L.item_id.get();
                if (!batched) {
                    k_item_list<<<grid_for((int64_t)Ic * nch), 256, 0, ls>>>(CA, mseg, mid);
                    FY_KERNEL_CHECK();
                }
                CA.item_seg = mseg;
                CA.item_id = mid;
                CA.item_grab = cooc_item_grab(((size_t)c < P.cluster_deg2.size() ? P.cluster_deg2[c] : P.sum_deg2) / (p.half ? 2 : 1), n_items);
                if (batched) batch_main.push_back(CoocLaunch{CA, ME, n_items, 0});
                else {
                    FY_HIP(hipMemsetAsync(L.any_overflow.get(), 0, sizeof(int32_t), ls));   // reused as the item counter
                    launch_cooc_rm2(ctx, tune, use_pk, CA, ME, n_items, L.any_overflow.get(), ls);
                }
            }
            t_cooc.end(sp, ls);
            join_values(ls);      // (the scoring kernels behind this point read the per-rating values)
            checkpoint("row kernel queued / run");
            if (!batched) R->st.cooc_launches++;
            if (p.half) {    // lower triangle + the block maxima in front of / on the diagonal
                const size_t sm = t_mirror.begin(ls);
                // pruned flow: LAZY mirror -- the seed pass reads the seed columns of every row, the bound pass the block maxima, the
                // survivor pass the surviving column blocks (mirrored below, once they are known); nothing else of the lower triangle
                // is ever read, so it is not written (round 3 moved 10.7 GB here to fill a triangle of which a few per cent were read)
                const bool lazy = p.prune && tune.lazy_mirror;
                launch_mirror(ctx, L.M.get(), ldm, Ic, p.prune ? L.Bmax.get() : nullptr, p.ldb, ls, nullptr, lazy ? std::min(tune.seed_chunks, p.nblk) : 0x7FFFFFFF);
                t_mirror.end(sm, ls);
            }
            // (not for long lists: their survivors -- 14 % of the blocks at N = 1000 -- spread over the whole popularity order, the 16 wide
            // groups behind the first 48 blocks all survive and hand every block to the survivor pass: measured, the batch falls back to
            // the full pass, 261 -> 471 ms)
            if (p.prune && !p.panel && tune.sup_bounds && !long_seed) {      // super-block bounds for the seed pass (k_score_sup)
                const int seed_b = std::min(tune.seed_chunks, p.nblk);
                std::vector<int32_t> first;
                sup_block_map(seed_b, p.nblk, first);
                const int n_sup = (int)first.size() - 1;
                first.resize(65, first.back());
                L.sup_first.alloc(ctx, 65);
                L.Bsup.alloc(ctx, (size_t)Ic * 64);
                L.asup.alloc(ctx, 64);
                L.bsupb.alloc(ctx, 64);
                FY_HIP(hipMemcpyAsync(L.sup_first.get(), first.data(), 65 * sizeof(int32_t), hipMemcpyHostToDevice, ls));
                FY_HIP(hipStreamSynchronize(ls));      // (`first` is a host temporary; one cluster: no other lane is waiting)
                const size_t sb = t_score.begin(ls);
                k_build_bsup<<<std::min<int>((Ic + 3) / 4, ctx->num_cus * 32), 256, 0, ls>>>(Ic, n_sup, L.sup_first.get(), L.Bmax.get(), p.ldb, L.Bsup.get());
                FY_KERNEL_CHECK();
                k_sup_amax<<<1, 64, 0, ls>>>(n_sup, L.sup_first.get(), L.amax.get(), L.bmax.get(), L.asup.get(), L.bsupb.get());
                FY_KERNEL_CHECK();
                t_score.end(sb, ls);
            }
            }      // do_build
            if (!do_score) continue;

            // -- scoring + top-N in user batches that fit the score scratch
            const int64_t ldS = ldm, B = p.B;
            const int n_chunks = (int)ceil_div(Ic, 64 * VEC);
            const int32_t* row_off = P.rowptr.get() + sbase;   // the users' CSR rows: [slot - sbase], [slot - sbase + 1]
            auto score_args = [&](const float* Mx, int64_t ldmx, int32_t Icx, const float* ax, int32_t s0, int32_t nb, float* Sx, int64_t ldSx,
                                  int n_slices, int nchunks) {
                ScoreArgs SA{};
                SA.M = Mx; SA.ldm = ldmx; SA.Ic = Icx; SA.a_rank = ax; SA.b_rank = b_rank32.get() + pbase; SA.rb_off = row_off;
                SA.csr_idx = P.csr_idx.get(); SA.csr_e = csr_e.get(); SA.csr_q = csr_q.get();
                SA.pvpi = pvpi.get(); SA.n_out = n_out.get(); SA.slot_lo = lo; SA.slot_base = sbase; SA.slot0 = s0; SA.n_users = nb;
                SA.S = Sx; SA.ldS = ldSx; SA.n_slices = n_slices; SA.n_chunks = nchunks;
                if (tune.score_heavy > 0) {     // (stream order: every scoring launch of the batch follows)
                    // a small batch (one cluster of many) cannot fill the chip with a wave per user and its launch lasts as long as its
                    // longest list on ONE wave (ML-1M shape in 50 clusters: 0.14 ms per cluster for 120 users): there every user with
                    // more than a few batches is walked by a whole workgroup
                    const int heavy = (int64_t)nb * nchunks < 8 * (int64_t)ctx->num_cus ? std::min(tune.score_heavy, 32) : tune.score_heavy;
                    k_count_heavy<<<1, 64, 0, ls>>>(P.rowptr.get() + s0, nb, heavy, L.n_heavy.get());
                    SA.n_heavy = L.n_heavy.get();
                }
                return SA;
            };
            // ill-conditioned list rows of a range of users, scored again in fp64 from the fp32 head rows (k_refine_rows); after the lists stand
            auto refine_rows = [&](int32_t s0, int32_t nb) {
                if (!refine || nb <= 0) return;
                const size_t tt = t_topn.begin(ls);
                const int32_t n_ids = J->S->max_item + 1;
                L.colmap.alloc(ctx, (size_t)n_ids);
                FY_HIP(hipMemsetAsync(L.colmap.get(), 0xFF, (size_t)n_ids * sizeof(int32_t), ls));
                k_refine_colmap<<<(head_rows + 255) / 256, 256, 0, ls>>>(head_rows, P.rank_item_raw.get() + pbase, L.colmap.get());
                FY_KERNEL_CHECK();
                RefineArgs RA{};
                RA.slot0 = s0; RA.n_users = nb; RA.slot_lo = lo; RA.slot_base = sbase;
                RA.n_out = n_out.get(); RA.out_off = out_off.get(); RA.out_item = R->d_key1.get(); RA.out_score = R->d_value.get(); RA.out_item_w = R->d_key1.get();
                RA.rowptr = P.rowptr.get(); RA.csr_idx = P.csr_idx.get(); RA.csr_r = P.csr_r.get(); RA.usum_slot = J->usum_slot.get();
                RA.p_rank = p_rank.get() + pbase; RA.b_rank = b_rank.get() + pbase;
                RA.colmap = L.colmap.get(); RA.max_item = J->S->max_item;
                RA.head_rows = head_rows;
                RA.head32 = H32.get(); RA.ld_head = ld_head; RA.tail32 = p.psym ? T32.get() : nullptr; RA.tail_from = p.p_eff;
                if (!RA.head32 || (p.psym && !RA.tail32)) FY_FAIL(FY_ERR_STATE, "internal: cluster %d has no fp32 rows for the refinement pass", c);
                RA.unscale = 1.0 / (double)h_gscale[(size_t)c];
                RA.lambda = lambda; RA.ln_items = std::log((double)prm.number_of_items); RA.ln_users = std::log((double)p.Uc);
                RA.users_minus_1 = (double)(p.Uc - 1);
                RA.refine_c = tune.refine_c;
                RA.n_refined = prune_counters.get() + 5;
                int lp2 = 64;
                while (lp2 < std::min<int>(prm.number_of_recommendations, Ic)) lp2 <<= 1;
                k_refine_rows<<<(nb + 3) / 4, 256, (size_t)4 * lp2 * sizeof(uint64_t), ls>>>(RA);
                FY_KERNEL_CHECK();
                t_topn.end(tt, ls);
            };
            // the plain full pass over a range of users: every log term, like the reference's loop (AbstractRM2Reducer.java:332-356)
            auto full_pass = [&](int32_t s0, int32_t nb, float* Sx) {
                const int n_slices = score_slices(ctx, tune, nb, n_chunks);
                ScoreArgs SA = score_args(L.M.get(), ldm, Ic, a_rank.get() + pbase, s0, nb, Sx, ldS, n_slices, n_chunks);
                const size_t ss = t_score.begin(ls);
                if (pack24) k_score<4, true, 8><<<n_chunks * n_slices, 256, 0, ls>>>(SA.M, SA.a_rank, SA.rb_off, SA.csr_idx, SA.csr_e, SA.csr_q, SA.pvpi, SA.n_out, SA.S, SA);
                else k_score<4, false, 8><<<n_chunks * n_slices, 256, 0, ls>>>(SA.M, SA.a_rank, SA.rb_off, SA.csr_idx, SA.csr_e, SA.csr_q, SA.pvpi, SA.n_out, SA.S, SA);
                FY_KERNEL_CHECK();
                t_score.end(ss, ls);
                R->st.score_launches++;
                TopNArgs TA{Sx, ldS, Ic, n_out.get(), out_off.get(), P.rank_item_raw.get() + pbase, P.slot2du.get(), P.uid.get(), lo, s0, c,
                            R->d_key0.get(), R->d_key1.get(), R->d_value.get(), R->d_aux.get(), 0, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr};
                const size_t tt = t_topn.begin(ls);
                FY_HIP(hipMemsetAsync(L.any_overflow.get(), 0, sizeof(int32_t), ls));
                if (prm.number_of_recommendations > TOPN_LONG)
                    k_topn_long<<<nb, 256, (size_t)fy_topn_long_cap(prm.number_of_recommendations) * 8, ls>>>(TA, L.overflow.get(), L.any_overflow.get(), tune.force_select,
                                                                                                         fy_topn_long_cap(prm.number_of_recommendations));
                else k_topn_fast<<<nb, 256, 0, ls>>>(TA, L.overflow.get(), L.any_overflow.get(), tune.force_select);
                FY_KERNEL_CHECK();
                k_topn_select<<<nb, 256, 0, ls>>>(TA, L.overflow.get(), L.any_overflow.get(), prune_counters.get() + 2);
                FY_KERNEL_CHECK();
                t_topn.end(tt, ls);
                refine_rows(s0, nb);
            };
            for (int32_t s0 = a; s0 < b; s0 += (int32_t)B) {
                const int32_t nb = (int32_t)std::min<int64_t>(B, b - s0);
                if (!p.prune) { full_pass(s0, nb, L.S.get()); continue; }
                const int seed_chunks = std::min(n_chunks, tune.seed_chunks);
                const bool use_sup = tune.sup_bounds && !p.panel && !long_seed;      // bounds over super-blocks, evaluated by the seed chunk's own waves
                const int n_slices = score_slices(ctx, tune, nb, seed_chunks + (use_sup ? 0 : (int)(p.ldb / 256)));
                // front in phase 2, back in phase 3 (the whole cluster in one batch: its CSR range is on the host already)
                const bool split = two_phase && p.panel && s0 == sbase && nb == p.Uc;
                if (phase == 3 && !split) continue;
                size_t ss = t_score.begin(ls);
                const int seed_blocks = seed_chunks;
                const int64_t SC = (int64_t)seed_chunks * 256;   // pitch of the compact score rows: seed columns only
                // only the seed columns and the surviving blocks of a score row are ever written or read
                // (1) + (3) ONE launch: exact scores of the seed columns (the most popular candidates) and the upper bounds
                // of all 256-column blocks (the same kernel on the block-maximum matrix); the grid's tail -- the waves
                // that walk the heaviest users -- is paid once instead of twice
                // panel mode: the stored rows are panel_cols wide and the bound matrix has one column per 64-column sub-block
                const float* Gmat = p.panel ? PP.Gp : L.M.get();
                const int64_t gld = p.panel ? (int64_t)p.panel_cols : ldm;
                // (A two-level bound -- 256-column block maxima first, the 64-column sub-block bounds only for the surviving blocks -- was
                // built and measured in round 3: the first level reads a quarter of the bytes, but five times as many blocks reach the
                // second level, whose per-(user, block) gathers of Bmax64 rows cost more than the streamed pass saved: 113 -> 129 ms at 50
                // clusters.  Removed.)
                const int64_t bld = p.panel ? p.ldb64 : p.ldb;           // pitch of the bound matrix, of UB and of the survivor lists
                const int bchunks = (int)(bld / 256);
                int32_t hv[3] = {0, 0, 0};   // survivors, first / last CSR entry of the batch
                if (phase != 3) {
                if (use_sup) {
                    ScoreArgs SU = score_args(Gmat, gld, Ic, a_rank.get() + pbase, s0, nb, L.S.get(), SC, n_slices, seed_chunks);
                    L.UBs.alloc(ctx, (size_t)nb * 64);
                    SU.Bsup = L.Bsup.get(); SU.asup = L.asup.get(); SU.bsup_b = L.bsupb.get(); SU.UBsup = L.UBs.get();
                    SU.n_sup = std::min(64, std::max(0, p.nblk - seed_chunks));
                    k_score_sup<<<seed_chunks * n_slices, 256, 0, ls>>>(SU.M, SU.a_rank, SU.rb_off, SU.csr_idx, SU.csr_e, SU.csr_q, SU.pvpi, SU.n_out, SU.S, SU);
                    FY_KERNEL_CHECK();
                }
                ScoreArgs SA = score_args(Gmat, gld, Ic, a_rank.get() + pbase, s0, nb, L.S.get(), SC, n_slices, seed_chunks + bchunks);
                SA.chunks1 = seed_chunks;
                SA.M2 = p.panel ? PP.Bmax64 : L.Bmax.get();
                SA.ldm2 = bld;
                SA.Ic2 = p.panel ? p.nsub : p.nblk;
                SA.a2 = p.panel ? PP.amax64 : L.amax.get();
                SA.b2 = p.panel ? PP.bmax64 : L.bmax.get();
                SA.S2 = L.UB.get();
                SA.ldS2 = bld;
                SA.no_mask2 = 1;
                if (!use_sup) {
                    k_score<4, true, 8><<<(seed_chunks + bchunks) * n_slices, 256, 0, ls>>>(SA.M, SA.a_rank, SA.rb_off, SA.csr_idx, SA.csr_e, SA.csr_q, SA.pvpi, SA.n_out, SA.S, SA);
                    FY_KERNEL_CHECK();
                }
                // (2) tau_u = N-th best seed score; the sorted seed head is also the user's list unless a block survives
                TopNArgs T1{L.S.get(), SC, Ic, n_out.get(), out_off.get(), P.rank_item_raw.get() + pbase, P.slot2du.get(), P.uid.get(),
                            lo, s0, c, R->d_key0.get(), R->d_key1.get(), R->d_value.get(), R->d_aux.get(),
                            1, seed_chunks * 256, L.surv.get(), L.n_quads.get(), bld, L.tau.get()};
                if (long_seed) {
                    k_topn_long<<<nb, 256, (size_t)fy_topn_long_cap(prm.number_of_recommendations) * 8, ls>>>(T1, L.overflow.get(), L.any_overflow.get(), 0,
                                                                                                         fy_topn_long_cap(prm.number_of_recommendations));
                } else {
                    int lp2 = 64;                   // the sort's size: the seed columns, at most TOPN_SAMPLE
                    while (lp2 < std::min<int>(std::min<int>(Ic, seed_chunks * 256), TOPN_SAMPLE)) lp2 <<= 1;
                    const unsigned sg = (unsigned)((nb + 3) / 4);
                    if (lp2 <= 64) k_topn_seed<1><<<sg, 256, 0, ls>>>(T1, nb, L.overflow.get());
                    else if (lp2 == 128) k_topn_seed<2><<<sg, 256, 0, ls>>>(T1, nb, L.overflow.get());
                    else if (lp2 == 256) k_topn_seed<4><<<sg, 256, 0, ls>>>(T1, nb, L.overflow.get());
                    else if (lp2 == 512) k_topn_seed<8><<<sg, 256, 0, ls>>>(T1, nb, L.overflow.get());
                    else k_topn_seed<16><<<sg, 256, 0, ls>>>(T1, nb, L.overflow.get());
                }
                FY_KERNEL_CHECK();
                // (4) the blocks whose bound reaches tau_u, in ascending order
                FY_HIP(hipMemsetAsync(L.n_quads.get(), 0, ((size_t)nb + 1) * sizeof(int32_t), ls));
                if (p.panel)
                    k_bound_select_sub<<<grid_for((int64_t)nb * 64, 256), 256, 0, ls>>>(L.UB.get(), bld, p.nsub, p.nblk, seed_blocks, L.tau.get(),
                                                                                       pvpi.get() + (s0 - lo), nb, bld, L.surv.get(), L.surv_mask.get(), L.n_quads.get());
                else if (use_sup)
                    k_bound_select_sup<<<grid_for((int64_t)nb * 64, 256), 256, 0, ls>>>(L.UBs.get(), std::min(64, std::max(0, p.nblk - seed_chunks)), L.sup_first.get(), L.tau.get(),
                                                                                       pvpi.get() + (s0 - lo), nb, p.ldb, L.surv.get(), L.n_quads.get());
                else
                    k_bound_select<<<grid_for((int64_t)nb * 64, 256), 256, 0, ls>>>(L.UB.get(), p.ldb, p.nblk, seed_blocks, L.tau.get(), pvpi.get() + (s0 - lo), nb,
                                                                                   L.surv.get(), L.n_quads.get());
                FY_KERNEL_CHECK();
                if (tune.debug_sync == 3 && !p.panel) {      // which blocks survive, and for how many users
                    DevBuf<int32_t> cnt(ctx, (size_t)p.nblk + 1);
                    FY_HIP(hipMemsetAsync(cnt.get(), 0, ((size_t)p.nblk + 1) * sizeof(int32_t), ls));
                    k_surv_block_counts<<<std::min<int>(nb, 4096), 64, 0, ls>>>(nb, L.n_quads.get(), L.surv.get(), p.ldb, cnt.get());
                    std::vector<int32_t> hc((size_t)p.nblk + 1);
                    FY_HIP(hipMemcpyAsync(hc.data(), cnt.get(), hc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ls));
                    FY_HIP(hipStreamSynchronize(ls));
                    int distinct = 0;
                    long long total = 0;
                    for (int b2 = 0; b2 < p.nblk; b2++) { distinct += hc[b2] > 0; total += hc[b2]; }
                    fprintf(stderr, "[fy] cluster %d: %lld surviving (user, block) pairs in %d distinct blocks of %d:", c, total, distinct, p.nblk);
                    for (int b2 = 0; b2 < p.nblk; b2++)
                        if (hc[b2]) fprintf(stderr, " %d:%d", b2, hc[b2]);
                    fprintf(stderr, "\n");
                }
                if (p.panel && tune.panel_repair) {
                    // (4b) sub-blocks that hold an item the user rated: bound again without the user's own co-ratings
                    RepairArgs RA{L.surv.get(), L.surv_mask.get(), L.n_quads.get(), bld, nb, s0, lo, p.p_eff, Ic, P.rowptr.get(), P.csr_idx.get(),
                                  csr_x.get(), csr_e.get(), csr_q.get(), PP.Bmax64, PP.Brep, p.ldb64, PP.amax64, PP.bmax64,
                                  L.tau.get(), pvpi.get(), (float)w2s, prune_counters.get()};
                    k_bound_repair<<<std::min<int>(nb, ctx->num_cus * 16), 256, 0, ls>>>(RA);
                    FY_KERNEL_CHECK();
                }
                if (p.half && !p.panel && tune.lazy_mirror) {      // lazy mirror, second pass: the column blocks with survivors
                    const size_t sm = t_mirror.begin(ls);
                    L.need.alloc(ctx, (size_t)p.nblk + 1);
                    FY_HIP(hipMemsetAsync(L.need.get(), 0, ((size_t)p.nblk + 1) * sizeof(int32_t), ls));
                    k_flag_surviving_blocks<<<std::min<int>(nb, 4096), 64, 0, ls>>>(nb, L.n_quads.get(), L.surv.get(), p.ldb, L.need.get());
                    FY_KERNEL_CHECK();
                    launch_mirror(ctx, L.M.get(), ldm, Ic, nullptr, p.ldb, ls, L.need.get(), 0, false);
                    t_mirror.end(sm, ls);
                }
                exclusive_scan_i32(ctx, L.n_quads.get(), L.quad_prefix.get(), (size_t)nb + 1, ls, &L.scan_tmp);
                if (split) {      // the count goes to pinned memory; the host does not wait here
                    FY_HIP(hipMemcpyAsync(&guard.pinned[pi], L.quad_prefix.get() + nb, sizeof(int32_t), hipMemcpyDeviceToHost, ls));
                    t_score.end(ss, ls);
                    checkpoint("front (seed + bound, select, second bound)");
                    continue;
                }
                }      // phase != 3
                // (5) exact scores of the survivors, packed: 256 floats per surviving block at entry quad_prefix[u] + k
                if (!split) FY_HIP(hipMemcpyAsync(&hv[0], L.quad_prefix.get() + nb, sizeof(int32_t), hipMemcpyDeviceToHost, ls));
                else hv[0] = guard.pinned[pi];
                if (s0 == sbase && nb == p.Uc) {      // the whole cluster: its CSR range is on the host already
                    hv[1] = csr_range[2 * pi];
                    hv[2] = csr_range[2 * pi + 1];
                } else {
                    FY_HIP(hipMemcpyAsync(&hv[1], P.rowptr.get() + s0, sizeof(int32_t), hipMemcpyDeviceToHost, ls));
                    FY_HIP(hipMemcpyAsync(&hv[2], P.rowptr.get() + s0 + nb, sizeof(int32_t), hipMemcpyDeviceToHost, ls));
                }
                if (!split) FY_HIP(hipStreamSynchronize(ls));   // (everything queued on this lane before has finished: Ssurv may be re-sized)
                const int32_t n_surv_total = hv[0];
                const int64_t blocks_checked = (int64_t)nb * std::max(0, p.nblk - seed_blocks);
                if (!p.panel && (double)n_surv_total > tune.max_surv_frac * (double)blocks_checked) {     // (panel mode has no full matrix to fall back on)
                    // The threshold did not bite (e.g. lambda = 0: a user who rated an item nobody else of the cluster rated has
                    // only -inf scores, tau = -inf keeps every block): the survivor pass would cost more than the plain full pass
                    // and 1 KB of scratch per survivor.  Redo the batch with the full pass, in sub-batches that fit the workspace.
                    t_score.end(ss, ls);
                    if (p.half && tune.lazy_mirror) {      // the plain full pass reads every row whole: the rest of the lower triangle now
                        const size_t sm = t_mirror.begin(ls);
                        launch_mirror(ctx, L.M.get(), ldm, Ic, nullptr, p.ldb, ls, nullptr, 0x7FFFFFFF, false);
                        t_mirror.end(sm, ls);
                    }
                    const int64_t sub = std::max<int64_t>(1, std::min<int64_t>((ws / NS) / (ldm * 4), nb));
                    DevBuf<float> Sfull(ctx, (size_t)(sub * ldm));
                    for (int32_t t0 = s0; t0 < s0 + nb; t0 += (int32_t)sub) full_pass(t0, (int32_t)std::min<int64_t>(sub, s0 + nb - t0), Sfull.get());
                    FY_HIP(hipStreamSynchronize(ls));   // Sfull goes back to the allocator
                    R->st.prune_fallbacks++;
                    R->st.score_launches++;          // the fused seed + bound launch that was thrown away
                    prune_blocks_total += blocks_checked;
                    fallback_survived += n_surv_total;
                    prune_seed_terms_cols += (int64_t)(hv[2] - hv[1]) * (seed_chunks * 256 + p.ldb + ldm);
                    continue;
                }
                L.Ssurv.alloc(ctx, (size_t)std::max(1, n_surv_total) * PRUNE_BLOCK);
                if (n_surv_total > 0) {
                    ScoreArgs SQ = score_args(Gmat, gld, Ic, a_rank.get() + pbase, s0, nb, L.Ssurv.get(), 0, n_slices, n_chunks);
                    const int panel_blocks = p.panel ? p.panel_cols / 256 : 0x7FFFFFFF;
                    k_score_blocks<8><<<std::min(n_surv_total, ctx->num_cus * 16), 256, 0, ls>>>(SQ.M, SQ.a_rank, P.rowptr.get(), SQ.csr_idx, SQ.csr_e, SQ.csr_q, SQ.pvpi,
                                                                        L.quad_prefix.get(), L.surv.get(), SQ.S, SQ, bld, prune_counters.get(), panel_blocks,
                                                                        p.panel ? L.surv_mask.get() : nullptr);
                    FY_KERNEL_CHECK();
                    if (p.panel && panel_blocks < p.nblk) {     // survivors behind the panel: exact, from the sparse data
                        const size_t slds = ((size_t)p.Uc + 1) * sizeof(int32_t);
                        const int sgrid = std::min<int>(n_surv_total, ctx->num_cus * (int)std::max<size_t>(1, std::min<size_t>(6, (150 * 1024) / (slds + 34 * 1024))));
                        L.strayT.alloc(ctx, (size_t)sgrid * STRAY_TCAP);
                        L.stray_items.alloc(ctx, (size_t)n_surv_total + 1);      // (a group holds at least one surviving block)
                        FY_HIP(hipMemsetAsync(L.any_overflow.get(), 0, sizeof(int32_t), ls));   // reused as the item counter
                        k_stray_items<<<grid_for(nb), 256, 0, ls>>>(L.quad_prefix.get(), L.surv.get(), bld, nb, panel_blocks, L.stray_items.get(),
                                                                   L.any_overflow.get());
                        FY_KERNEL_CHECK();
                        StrayArgs ST{L.quad_prefix.get(), L.surv.get(), L.surv_mask.get(), bld, nb, s0, lo, sbase, p.Uc, panel_blocks, Ic, pbase,
                                     P.rowptr.get(), P.csr_idx.get(), csr_e.get(), csr_q.get(), P.rank_pair.get(), P.pair_start.get(), P.csc_slot.get(),
                                     csc_x.get(), a_rank.get() + pbase, b_rank32.get() + pbase, pvpi.get(),
                                     (float)w2s, L.Ssurv.get(), L.strayT.get(), prune_counters.get(), L.stray_items.get(),
                                     L.any_overflow.get()};
                        k_score_stray<<<sgrid, 256, slds, ls>>>(ST);
                        FY_KERNEL_CHECK();
                    }
                }
                t_score.end(ss, ls);
                R->st.score_launches += 2;   // the fused seed + bound launch and the survivor pass
                prune_blocks_total += blocks_checked;
                // log terms of the seed and bound passes of this batch: (ratings of its users) x (columns walked)
                prune_seed_terms_cols += (int64_t)(hv[2] - hv[1]) * (seed_chunks * 256 + (use_sup ? 64 : bld));
                const int32_t seed_cols_p = seed_chunks * 256;
                TopNArgs TA{L.S.get(), (int64_t)seed_cols_p, Ic, n_out.get(), out_off.get(), P.rank_item_raw.get() + pbase,
                            P.slot2du.get(), P.uid.get(), lo, s0, c, R->d_key0.get(), R->d_key1.get(), R->d_value.get(), R->d_aux.get(),
                            2, seed_cols_p, L.surv.get(), L.n_quads.get(), bld, L.tau.get(), L.Ssurv.get(), L.quad_prefix.get()};
                const size_t tt = t_topn.begin(ls);
                FY_HIP(hipMemsetAsync(L.any_overflow.get(), 0, sizeof(int32_t), ls));
                if (long_seed)
                    k_topn_long<<<nb, 256, (size_t)fy_topn_long_cap(prm.number_of_recommendations) * 8, ls>>>(TA, L.overflow.get(), L.any_overflow.get(), tune.force_select,
                                                                                                         fy_topn_long_cap(prm.number_of_recommendations));
                else k_topn_fast<<<nb, 256, 0, ls>>>(TA, L.overflow.get(), L.any_overflow.get(), tune.force_select);
                FY_KERNEL_CHECK();
                k_topn_select<<<nb, 256, 0, ls>>>(TA, L.overflow.get(), L.any_overflow.get(), prune_counters.get() + 2);
                FY_KERNEL_CHECK();
                t_topn.end(tt, ls);
                refine_rows(s0, nb);
                checkpoint("back (survivors, strays, lists)");
            }
        }
        if (phase == 1 && (!batch_main.empty() || !batch_tail.empty())) {     // the row kernels of all panel clusters: two launches
            const size_t sp = t_cooc.begin(st);
            launch_cooc_rm2_multi(ctx, tune, batch_tail, true, d_batch_tail, d_cnt_tail, st, true);
            launch_cooc_rm2_multi(ctx, tune, batch_main, false, d_batch_main, d_cnt_main, st, true);
            t_cooc.end(sp, st);
            // symmetric panel mode: the lower triangle of every panel's square, then the head rows' bounds (three launches for all clusters)
            int max_cols = 0, max_nsub = 0;
            for (size_t pi = 0; pi < plans.size(); pi++) {
                const Plan& p = plans[pi];
                if (!p.psym || group_of[pi] != grp) continue;
                hpd.push_back(PanelDesc{pbuf[pi].Gp.get(), pbuf[pi].Bmax64.get(), pbuf[pi].Brep.get(), p.panel_cols, p.ldb64, p.Ic, p.p_eff, p.nsub, 0});
                max_cols = std::max(max_cols, p.panel_cols);
                max_nsub = std::max(max_nsub, p.nsub);
            }
            if (!hpd.empty()) {
                // operand check on the host (round 3: these launches faulted at address 0x1000 when a half-built group handed them the
                // descriptors of clusters whose panels were not allocated): every panel pointer set, every shape what the kernels' grids assume
                for (size_t k = 0; k < hpd.size(); k++) {
                    const PanelDesc& d = hpd[k];
                    if (!d.Gp || !d.Bmax64 || !d.Brep || d.panel_cols <= 0 || d.panel_cols % 256 != 0 || d.p_eff != d.panel_cols || d.Ic < d.p_eff || d.nsub <= 0 ||
                        d.ldb64 < d.nsub)
                        FY_FAIL(FY_ERR_STATE, "internal: panel %zu of %zu of a batched mirror / column-maximum launch is unusable (Gp %p Bmax64 %p Brep %p panel_cols %d p_eff %d Ic %d nsub %d ldb64 %lld)",
                                k, hpd.size(), (const void*)d.Gp, (const void*)d.Bmax64, (const void*)d.Brep, (int)d.panel_cols, (int)d.p_eff, (int)d.Ic, (int)d.nsub, (long long)d.ldb64);
                }
                const size_t sm = t_mirror.begin(st);
                d_panel_desc.alloc(ctx, hpd.size());
                FY_HIP(hipMemcpyAsync(d_panel_desc.get(), hpd.data(), hpd.size() * sizeof(PanelDesc), hipMemcpyHostToDevice, st));
                const unsigned nz = (unsigned)hpd.size();
                if (max_cols / 256 > 1) {
                    FY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mirror_tiles_multi), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * MIRROR_PITCH));
                    k_mirror_tiles_multi<<<dim3((unsigned)(max_cols / 128), (unsigned)(max_cols / 256 - 1), nz), 1024, 128 * MIRROR_PITCH, st>>>(d_panel_desc.get());
                    FY_KERNEL_CHECK();
                }
                k_mirror_diag_multi<<<dim3((unsigned)(max_cols / 256), nz), 256, 0, st>>>(d_panel_desc.get());
                FY_KERNEL_CHECK();
                k_panel_colmax<<<dim3((unsigned)(max_cols / 256), (unsigned)ceil_div(max_nsub, 16), nz), 256, 0, st>>>(d_panel_desc.get());
                FY_KERNEL_CHECK();
                t_mirror.end(sm, st);
            }
            if (tune.debug_sync == 1) {
                const hipError_t e = hipDeviceSynchronize();
                fprintf(stderr, "[fy] group %d/%d: batched row kernels, mirror, column maxima -> %s\n", grp, n_groups, hipGetErrorString(e));
                fflush(stderr);
            }
            R->st.cooc_launches += (batch_tail.empty() ? 0 : 1) + 1;
        }
        if (tune.debug_sync == 2) {
            (void)hipDeviceSynchronize();
            fprintf(stderr, "[fy] group %d phase %d: %.3f ms from its first queued operation to the drained device\n", grp, phase,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count());
        }
        }      // phase
        if (grp + 1 < n_groups) {     // the next group re-uses this group's memory: everything queued so far must have finished
            for (int l = 0; l < NS; l++) FY_HIP(hipStreamSynchronize(lanes[l].st));
            FY_HIP(hipStreamSynchronize(st));
        }
        }      // group
        if (NS > 1) {   // join: the main stream continues after every lane has drained
            for (int l = 0; l < NS; l++) {
                hipEvent_t done;
                FY_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
                FY_HIP(hipEventRecord(done, lanes[l].st));
                FY_HIP(hipStreamWaitEvent(st, done, 0));
                FY_HIP(hipEventDestroy(done));
            }
        }
        // (the guard's destructor drains the lanes and the main stream before any buffer of this scope is released)
        {
            unsigned long long hc[6];
            d2h(ctx, hc, prune_counters.get(), 6);
            sync(ctx);
            if (!tables_cached && !all_at_once) {        // (build_tables_all counts its one table itself)
                tc.total_segments = 0;
                for (auto& t : segs) tc.total_segments += t.n_seg;
                for (auto& t : segs_tail) tc.total_segments += t.n_seg;
            }
            tc.sig = sig;          // every table of the plan has been built and used: a later job with the same plan re-uses them
            tc.valid = true;
            R->st.cooc_segments = tc.total_segments;
            for (auto& p : plans) {       // what the row kernels store (the mirror pass and the column maxima are priced separately)
                const int64_t eb = p.pack24 ? 3 : 4;
                if (p.coop) continue;
                if (p.panel) R->st.cooc_matrix_bytes += (int64_t)p.Ic * p.panel_cols * 3 + (int64_t)p.Ic * p.ldb64 * 7;
                else R->st.cooc_matrix_bytes += eb * (p.half ? (int64_t)p.Ic * (p.Ic + 256) / 2 : (int64_t)p.Ic * p.Ic) + (p.prune ? (int64_t)p.Ic * p.nblk * 3 : 0);
            }
            R->st.topn_select_users = (int64_t)hc[2];
            R->st.stray_blocks = (int64_t)hc[3];
            R->st.bound_repairs = (int64_t)hc[4];
            R->st.rows_refined = (int64_t)hc[5];
            R->st.blocks_survived = (int64_t)hc[0] + coop_survived + fallback_survived;
            R->st.blocks_total = prune_blocks_total;
            R->st.log_terms_evaluated = prune_blocks_total ? (int64_t)hc[1] + prune_seed_terms_cols : 0;
        }
    }
    // rm2/userSum and rm2/itemColl stay in HBM until somebody asks for them
    if (J->S->sharded) {
        // the GLOBAL statistics (jobs RM2-1 / RM2-2 write one userSum and one itemColl for all clusters): out of the summed raw-id buffer
        const RM2Static& S = *J->S;
        const double* raw_items = J->stats_raw.get();
        const double* raw_users = J->stats_raw.get() + S.n_raw_items + 1;
        DevBuf<uint32_t> fi(ctx, (size_t)S.n_raw_items), pi(ctx, (size_t)S.n_raw_items), fu(ctx, (size_t)S.n_raw_users), pu(ctx, (size_t)S.n_raw_users);
        k_flag_positive<<<grid_for(S.n_raw_items), 256, 0, st>>>(S.n_raw_items, raw_items, fi.get());
        FY_KERNEL_CHECK();
        k_flag_positive<<<grid_for(S.n_raw_users), 256, 0, st>>>(S.n_raw_users, raw_users, fu.get());
        FY_KERNEL_CHECK();
        inclusive_scan_u32(ctx, fi.get(), pi.get(), (size_t)S.n_raw_items);
        inclusive_scan_u32(ctx, fu.get(), pu.get(), (size_t)S.n_raw_users);
        uint32_t n_items_all = 0, n_users_all = 0;
        d2h(ctx, &n_items_all, pi.get() + (S.n_raw_items - 1), 1);
        d2h(ctx, &n_users_all, pu.get() + (S.n_raw_users - 1), 1);
        sync(ctx);
        R->d_item_id.alloc(ctx, n_items_all);
        R->d_icoll.alloc(ctx, n_items_all);
        R->d_user_id.alloc(ctx, n_users_all);
        R->d_user_sum.alloc(ctx, n_users_all);
        k_compact_positive<<<grid_for(S.n_raw_items), 256, 0, st>>>(S.n_raw_items, raw_items, pi.get(), d_total.get(), R->d_item_id.get(), R->d_icoll.get());
        FY_KERNEL_CHECK();
        k_compact_positive<<<grid_for(S.n_raw_users), 256, 0, st>>>(S.n_raw_users, raw_users, pu.get(), nullptr, R->d_user_id.get(), R->d_user_sum.get());
        FY_KERNEL_CHECK();
        R->st.n_users = n_users_all;
        R->st.n_items = n_items_all;
        sync(ctx);      // (the flags and positions are released here)
    } else {
        R->d_user_id.alloc(ctx, nU);
        R->d_user_sum.alloc(ctx, nU);
        d2d(ctx, R->d_user_id.get(), P.uid.get(), nU);
        d2d(ctx, R->d_user_sum.get(), P.usum.get(), nU);
        R->d_item_id.alloc(ctx, nI);
        d2d(ctx, R->d_item_id.get(), P.iid.get(), nI);
    }
    t_total.end(span_total);
    d2h(ctx, &R->total_sum, d_total.get(), 1);
    sync(ctx);
    // (bench.py prices the row kernel with (pair_contribs - nnz) / 2 unordered pairs: a cooperative rank walked only its rows)
    R->st.pair_contribs = any_coop ? coop_pair_contribs + P.nnz : P.sum_deg2;
    R->st.ms_cooc = t_cooc.total_ms();
    R->st.ms_score = t_score.total_ms();
    R->st.ms_topn = t_topn.total_ms();
    R->st.ms_total = t_total.total_ms();
    R->st.ms_tables = t_tables.total_ms();
    R->st.ms_mirror = t_mirror.total_ms();
    return R.release();
}

void fy::rm2_set_collectives(fy_rm2_job* J, const fy_collectives* c) {
    if (!c || !c->all_gather || !c->reduce_scatter_f32) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "fy_collectives needs all_gather and reduce_scatter_f32");
    J->coll = *c;
    J->have_coll = true;
}

void fy::rm2_partial_stats(fy_rm2_job* J, double** buf, int64_t* len) {
    *buf = J->S->sharded ? J->S->partial_raw.get() : J->partial.get();
    *len = J->S->exchange_len();
}

void fy::rm2_stats_layout(fy_rm2_job* J, int64_t* n_item_slots, int64_t* n_user_slots) {
    *n_item_slots = J->S->sharded ? J->S->n_raw_items : (int64_t)J->P.nI;
    *n_user_slots = J->S->sharded ? J->S->n_raw_users : 0;
}

void fy::rm2_job_destroy(fy_rm2_job* J) {
    if (!J) return;
    Context* ctx = J->ctx;
    // nothing of the job may still be read by a queued kernel when its arrays go back to the caching allocator
    for (hipStream_t x : ctx->aux) (void)hipStreamSynchronize(x);
    (void)hipStreamSynchronize(ctx->stream);
    delete J;
}
