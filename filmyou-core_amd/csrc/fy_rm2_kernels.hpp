// fy_rm2_kernels.hpp -- device code of the RM2 job's scoring side: the scoring kernel family, top-N, the branch-and-bound
// helpers and the kernels of the cooperative multi-rank path.  NOT a standalone header: it is one section of fy_rm2.hip's
// translation unit (included inside `namespace fy`, after the statistics / row-kernel sections), split out for size.
#pragma once

// ================================================================ scoring kernel
// Work item = (user, column chunk of 64*VEC items): the wave walks the user's CSR row (wave-uniform scalar loads of idx,
// e and q) and for every rated item j streams the segment G[j][chunk] (4*VEC bytes per lane, coalesced), adding
// log2(G[j][i] + q_j * b_i + a_i * e_uj) to the lane's VEC candidates (G = (1-l)^2 X^T X, q_j = l (1-l) p_j, a_i = l p_i:
// the rank-one part of the reference's inner sum is applied here, one FMA per term, so that the stored matrix is the pure
// co-rating Gram -- symmetric, and zero wherever two items were never co-rated).  No cross-lane traffic at all.
// (Round 1 also measured a row-blocked variant sized to the L2, a variant with the hottest rows resident in LDS and
// non-temporal loads of the cold rows: all slower; they are no longer in the tree.)
struct ScoreArgs {
    const float* __restrict__ M;
    int64_t ldm;
    int32_t Ic;
    const float* __restrict__ a_rank;      // l * p_i in rank order, offset by pbase
    const float* __restrict__ b_rank;      // b_i = sum_v x_vi (fp32) in rank order, offset by pbase
    const int32_t* __restrict__ rb_off;    // [slot - slot_base], [slot - slot_base + 1]: the user's CSR row (rowptr of the cluster)
    const int32_t* __restrict__ csr_idx;
    const float* __restrict__ csr_e;
    const float* __restrict__ csr_q;       // q_j = l (1-l) p_j per rating (same indexing as csr_e)
    const double* __restrict__ pvpi;       // indexed by slot - slot_lo
    const int32_t* __restrict__ n_out;     // indexed by slot - slot_lo; 0 = user gets no list
    int32_t slot_lo;                       // first slot of this rank
    int32_t slot_base;                     // first slot of the cluster (rb_off origin)
    int32_t slot0;                         // first slot of this batch
    int32_t n_users;                       // users in the batch
    float* __restrict__ S;                 // [n_users][ldS]
    int64_t ldS;
    int32_t n_slices;
    int32_t n_chunks;
    int32_t no_mask;                       // != 0: the columns are not items (bound pass over block maxima): no "already rated" mask
    // cooperative ranks: csr_idx holds LOCAL row indices of M (k-th row of the rank); the item it stands for is
    // k * row_mul + row_add (row_mul == 0: the index is the item itself)
    int32_t row_mul, row_add;
    // fused launch: chunks [chunks1, n) of the grid work on a second matrix (the block maxima), results stored like scores
    int32_t chunks1;                       // 0 = single matrix
    const float* __restrict__ M2;
    int64_t ldm2;
    int32_t Ic2;
    const float* __restrict__ a2;
    const float* __restrict__ b2;
    float* __restrict__ S2;
    int64_t ldS2;
    int32_t no_mask2;
    // the first *n_heavy users of the batch (slots are in descending order of degree) are walked by the four waves of a
    // workgroup together, a quarter of the list each; nullptr = none
    const int32_t* __restrict__ n_heavy;
    // super-block bounds riding with the seed chunk (score_body<..., SUP = true>, one-cluster pruned job): Bsup[j][s], s < 64 = the
    // largest matrix entry of row j inside super-block s (fp32, one dword per lane), asup / bsup_b the super-blocks' maxima of a / b,
    // UBsup[u][64] receives the bounds (NaN for s >= n_sup)
    const float* __restrict__ Bsup;
    const float* __restrict__ asup;
    const float* __restrict__ bsup_b;
    float* __restrict__ UBsup;
    int32_t n_sup;
};
// number of users at the head of a batch (descending degree) with more than `thresh` ratings
__global__ void k_count_heavy(const int32_t* __restrict__ rowptr /* of the batch's first slot */, int32_t n_users, int32_t thresh,
                              int32_t* __restrict__ out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int lo = 0, hi = n_users;       // first u with degree <= thresh
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rowptr[mid + 1] - rowptr[mid] > thresh) lo = mid + 1; else hi = mid; }
        *out = lo;
    }
}

__device__ __forceinline__ float fy_log2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32
// Sum of log2 over a batch of terms WITHOUT one transcendental per term (round 4): x = m * 2^e with m in [0.5, 1)
// (v_frexp_mant_f32 / v_frexp_exp_i32_f32, full-rate VALU), the mantissas of a batch are MULTIPLIED in fp32 (eight factors: the
// product stays above 2^-8, relative error <= 7 * 2^-24), the exponents are added as integers (exact), and the batch costs ONE
// v_log_f32 of the product: sum log2 x = log2(prod m) + sum e.  Against one v_log_f32 per term this is (a) more exact -- a log2 of
// -13 .. -40 carried in one fp32 has an ulp of 1e-6 .. 4e-6, which neither v_log_f32's result nor fp32 sums of such values can beat,
// while log2 of a product in [2^-8, 1) resolves 5e-7 for the whole batch: the all-rows error against the fp64 definition at ML-25M
// shape fell from 4.7e-6 / 7.9e-6 (one / 50 clusters) to 3.5e-6 / 5.8e-6 with the split alone -- and (b) cheaper: four full-rate
// instructions per term and half a transcendental cycle instead of a quarter-rate v_log_f32 (four cycles) and an add.
// 0 -> mantissa 0: the product is 0 and its log2 -inf, as before (U_c = 1, quirk Q7); +inf and NaN pass through the mantissa.
__device__ __forceinline__ void fy_logprod_step(float x, float& mant_prod, int& exp_sum) {
    mant_prod *= __builtin_amdgcn_frexp_mantf(x);
    exp_sum += __builtin_amdgcn_frexp_expf(x);
}
__device__ __forceinline__ double fy_logprod_fold(float mant_prod, int exp_sum) {      // log2 of the batch's product, fp64
    return (double)__builtin_amdgcn_logf(mant_prod) + (double)exp_sum;
}

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

// The read-only arrays are separate `const T* __restrict__` kernel arguments (not struct members): only then does hipcc
// prove them invariant and fetch the wave-uniform idx / e / offsets with scalar loads (s_load) instead of a vector
// load + v_readfirstlane in front of every row-segment load.
struct U3 {
    uint32_t a, b, c;
};
// four packed 24-bit values (12 bytes; 7 exponent + 17 mantissa bits of a float < 2, no sign: FY_P24_SHIFT in fy_rm2.hip) -> four
// floats: v_perm_b32 (the three bytes into the upper three of a dword) + shift each
__device__ __forceinline__ void fy_unpack24(const U3& d, float* f) {
    f[0] = __uint_as_float(__builtin_amdgcn_perm(0u, d.a, 0x0201000cu) >> (8 - FY_P24_SHIFT));
    f[1] = __uint_as_float(__builtin_amdgcn_perm(d.b, d.a, 0x0504030cu) >> (8 - FY_P24_SHIFT));
    f[2] = __uint_as_float(__builtin_amdgcn_perm(d.c, d.b, 0x0403020cu) >> (8 - FY_P24_SHIFT));
    f[3] = __uint_as_float(__builtin_amdgcn_perm(0u, d.c, 0x0302010cu) >> (8 - FY_P24_SHIFT));
}

// Does a block with upper bound `ub` have to be scored exactly for a user whose N-th best seed score is `tau`?
// tau = +inf: the user emits nothing; tau = -inf: fewer than N finite seed scores, nothing can be excluded (ties at -inf
// are broken by item id, so even a block of -inf scores may contribute).
// The margin covers the fp32 rounding of BOTH sums (the bound and the seed score behind tau).  That rounding scales with
// the magnitude of the summed log terms, |ub - pvpi|, not with the net value |ub|: pvpi is positive whenever
// U_c < numberOfItems (every multi-cluster job) and cancels most of the negative log sum.  Error model: a log2 term is
// ~ -13 .. -40, eight of them are added in fp32 (partial sums up to ~300: ulp 3e-5, at most 8 roundings = 1.2e-4 per
// batch, i.e. <= 1e-6 relative to the batch), the batches are folded in fp64: |error| <= 1e-6 * sum |log| in the worst
// case.  With S = |pvpi| + |ub - pvpi| >= sum |log| the margin 4e-6 * S + 1e-4 is twice that bound for each of the two sums.
__device__ __forceinline__ bool fy_bound_keeps(float ub, float tau, float pvpi) {
    if (!(ub == ub) || tau == INFINITY) return false;
    if (tau == -INFINITY) return true;
    const float S = fabsf(pvpi) + fabsf(ub - pvpi);
    return ub + (4e-6f * S + 1e-4f) >= tau;
}

template <int VEC, bool P24, int SB, bool SUP = false>
__device__ __forceinline__ void score_body(const float* __restrict__ M_, const float* __restrict__ a_rank_,
                                           const int32_t* __restrict__ rb_off_, const int32_t* __restrict__ csr_idx_,
                                           const float* __restrict__ csr_e_, const float* __restrict__ csr_q_,
                                           const double* __restrict__ pvpi_,
                                           const int32_t* __restrict__ n_out_, float* __restrict__ S_, const ScoreArgs& A, const int bx /* work-group index */) {
    using V = typename VecT<VEC>::type;
    using G = typename std::conditional<P24, U3, V>::type;   // what one lane loads per row
    static_assert(!P24 || VEC == 4, "24-bit rows are packed four columns to three dwords");
    constexpr int CW = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int chunk = bx / A.n_slices;
    const int slice = bx - chunk * A.n_slices;
    // fused seed + bound launch (one grid, one tail): the chunks behind the first `chunks1` belong to a second matrix
    const bool second = A.chunks1 > 0 && chunk >= A.chunks1;
    if (second) chunk -= A.chunks1;
    const float* __restrict__ Msel = second ? A.M2 : M_;
    const float* __restrict__ asel = second ? A.a2 : a_rank_;
    const float* __restrict__ bsel = second ? A.b2 : A.b_rank;
    float* __restrict__ Ssel = second ? A.S2 : S_;
    const int64_t ldm_sel = second ? A.ldm2 : A.ldm, ldS_sel = second ? A.ldS2 : A.ldS;
    const int Ic_sel = second ? A.Ic2 : A.Ic, no_mask = second ? A.no_mask2 : A.no_mask;
    const int col0 = chunk * CW;
    const int col = col0 + lane * VEC;
    float a[VEC], bb[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        a[v] = col + v < Ic_sel ? asel[col + v] : 0.0f;
        bb[v] = col + v < Ic_sel ? bsel[col + v] : 0.0f;
    }
    // SUP: the wave that scores seed chunk 0 also evaluates the user's bounds of up to 64 SUPER-BLOCKS (lane = super-block): one more
    // dword per lane and row, one more log term per lane -- instead of a second work item per user that streams a 768-byte row of
    // block maxima per rated item (round 3: seed chunk + bound chunk = 1536 B per rating; now 768 + 256)
    const bool sup_here = SUP && chunk == 0 && !second;
    float as_ = 0.f, bs_ = 0.f;
    if constexpr (SUP) {
        if (sup_here && lane < A.n_sup) { as_ = A.asup[lane]; bs_ = A.bsup_b[lane]; }
    }
    const float* __restrict__ Bsup_lane = SUP ? A.Bsup + lane : nullptr;
    // byte address of this lane's part of row 0; the row pitch is ldm * (P24 ? 3 : 4) bytes
    const char* __restrict__ Mcol = reinterpret_cast<const char*>(Msel) + (int64_t)col * (P24 ? 3 : 4);
    const int64_t pitch = ldm_sel * (P24 ? 3 : 4);
    const double LN2 = 0.69314718055994530942;
    const float qnan = __builtin_nanf("");
    const int row_mul = A.row_mul ? A.row_mul : 1;
    // the log terms of the rows [beg, end) of one user's list, added to t / mask
    // Batches of SB rows.  The (idx, e, q) triplets of a batch are fetched by THREE vector loads (lane l < SB holds row k + l) one
    // batch ahead and handed out with v_readlane.  (Until round 3 they were wave-uniform loads: the compiler issued the SB index
    // loads one at a time, each behind its own s_waitcnt lgkmcnt(0), and two more scalar loads per row for e and q in front of its
    // logs -- 24 dependent scalar round trips per batch of 8 rows, which is what "bound by scalar-load latency" meant.)
    // All SB row-segment loads are issued before the first use, also for a short tail (out-of-range slots re-load the last valid
    // row -- an L1 hit -- and are skipped by a wave-uniform test).
    auto walk = [&](int beg, int end, double* t, unsigned& mask, double& ts) __attribute__((always_inline)) {
        if (beg >= end) return;
        static_assert(SB <= 64, "one lane per row of a batch");
        const int sub = lane & (SB - 1);     // (SB is a power of two)
        int vi, vi_n;
        float ve, vq, ve_n, vq_n;
        {
            const int kk = min(beg + sub, end - 1);
            vi = csr_idx_[kk]; ve = csr_e_[kk]; vq = csr_q_[kk];
        }
        for (int k = beg; k < end; k += SB) {
            G g[SB];
            float e[SB], qq[SB];
            int jj[SB];
            float gs[SUP ? SB : 1];
#pragma unroll
            for (int q = 0; q < SB; q++) {
                jj[q] = __builtin_amdgcn_readlane(vi, q);
                e[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ve), q));
                qq[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vq), q));
                g[q] = *reinterpret_cast<const G*>(Mcol + (int64_t)jj[q] * pitch);
                if constexpr (SUP) {
                    if (sup_here) gs[q] = Bsup_lane[(int64_t)jj[q] * 64];      // (wave-uniform branch)
                }
            }
            {   // the next batch's triplets (clamped: the last batch re-reads the list's last row)
                const int kk = min(k + SB + sub, end - 1);
                vi_n = csr_idx_[kk]; ve_n = csr_e_[kk]; vq_n = csr_q_[kk];
            }
            // per batch: the fp32 PRODUCT of the SB mantissas and the exact integer sum of the exponents, one v_log_f32 per batch
            // (fy_logprod_step / fy_logprod_fold), folded into fp64.  (Round 3 summed whole log2 values, four at a time, in fp32:
            // partial sums up to ~160, ulp 1.5e-5 -- for users whose log sum nearly cancels pvpi that noise was as large as the matrix
            // format's rounding.)
            float p[VEC];
            int pe[VEC];
#pragma unroll
            for (int v = 0; v < VEC; v++) { p[v] = 1.f; pe[v] = 0; }
            float ps = 1.f;
            int pes = 0;
#pragma unroll
            for (int q = 0; q < SB; q++) {
                if (k + q < end) {
                    if constexpr (SUP) {
                        if (sup_here) fy_logprod_step(fmaf(qq[q], bs_, fmaf(as_, e[q], gs[q])), ps, pes);
                    }
                    float gv[VEC];
                    if constexpr (P24) fy_unpack24(g[q], gv);
                    else {
                        const float* gp = reinterpret_cast<const float*>(&g[q]);
#pragma unroll
                        for (int v = 0; v < VEC; v++) gv[v] = gp[v];
                    }
#pragma unroll
                    for (int v = 0; v < VEC; v++) fy_logprod_step(fmaf(qq[q], bb[v], fmaf(a[v], e[q], gv[v])), p[v], pe[v]);
                    const unsigned d = (unsigned)(jj[q] * row_mul + A.row_add - col0);
                    if (!no_mask && d < (unsigned)CW && (int)(d / VEC) == lane) mask |= 1u << (d % VEC);
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; v++) t[v] += fy_logprod_fold(p[v], pe[v]);
            if constexpr (SUP) {
                if (sup_here) ts += fy_logprod_fold(ps, pes);
            }
            vi = vi_n; ve = ve_n; vq = vq_n;
        }
    };
    auto store = [&](int u, int slot, const double* t, unsigned mask) __attribute__((always_inline)) {
        V o;
        float* ov = reinterpret_cast<float*>(&o);
        const double base = pvpi_[slot - A.slot_lo];
#pragma unroll
        for (int v = 0; v < VEC; v++) ov[v] = (float)(base + LN2 * t[v]);
#pragma unroll
        for (int v = 0; v < VEC; v++)
            if ((mask >> v) & 1u || col + v >= Ic_sel) ov[v] = qnan;
        *reinterpret_cast<V*>(Ssel + (int64_t)u * ldS_sel + col) = o;
    };
    auto store_sup = [&](int u, int slot, double ts) __attribute__((always_inline)) {
        if constexpr (SUP) {
            if (sup_here) A.UBsup[(int64_t)u * 64 + lane] = lane < A.n_sup ? (float)(pvpi_[slot - A.slot_lo] + LN2 * ts) : qnan;
        }
    };
    // ---- heavy users (the head of the batch): one per workgroup, a quarter of the list per wave.  A user with 3 000 ratings on
    // ONE wave (375 batches, one round trip each) was what a cluster's launch waited for: 1.1 ms for a 3 000-user cluster
    // whose average wave was done after 0.1 ms.
    __shared__ double sh_t[3][64][VEC];
    __shared__ double sh_ts[SUP ? 3 : 1][64];
    __shared__ unsigned sh_mask[3][64];
    const int n_heavy = A.n_heavy ? min(*A.n_heavy, A.n_users) : 0;
    for (int u = slice; u < n_heavy; u += A.n_slices) {            // block-uniform
        const int slot = A.slot0 + u;
        if (n_out_[slot - A.slot_lo] == 0) continue;
        const int32_t* __restrict__ ro = rb_off_ + (slot - A.slot_base);
        const int beg = ro[0], end = ro[1];
        const int quarter = (((end - beg + 3) >> 2) + SB - 1) / SB * SB;   // whole batches per wave
        const int wb = min(end, beg + wave * quarter), we = min(end, wb + quarter);
        double t[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) t[v] = 0.0;
        unsigned mask = 0;
        double ts = 0.0;
        walk(wb, we, t, mask, ts);
        if (wave > 0) {
#pragma unroll
            for (int v = 0; v < VEC; v++) sh_t[wave - 1][lane][v] = t[v];
            sh_mask[wave - 1][lane] = mask;
            if constexpr (SUP) sh_ts[wave - 1][lane] = ts;
        }
        __syncthreads();
        if (wave == 0) {
            for (int x = 0; x < 3; x++) {
#pragma unroll
                for (int v = 0; v < VEC; v++) t[v] += sh_t[x][lane][v];
                mask |= sh_mask[x][lane];
                if constexpr (SUP) ts += sh_ts[x][lane];
            }
            store(u, slot, t, mask);
            store_sup(u, slot, ts);
        }
        __syncthreads();   // sh_t is free again
    }
    for (int u = n_heavy + slice * 4 + wave; u < A.n_users; u += A.n_slices * 4) {
        const int slot = A.slot0 + u;
        if (n_out_[slot - A.slot_lo] == 0) continue;
        const int32_t* __restrict__ ro = rb_off_ + (slot - A.slot_base);
        const int beg = ro[0], end = ro[1];
        double t[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) t[v] = 0.0;
        unsigned mask = 0;
        double ts = 0.0;
        walk(beg, end, t, mask, ts);
        store(u, slot, t, mask);
        store_sup(u, slot, ts);
    }
}
// (Round 4 held the register allocator to 64 VGPRs -- eight waves per SIMD instead of the seven that its 68 allow -- with
// amdgpu_waves_per_eu(8, 8): 6.19 against 6.14 ms for the scoring family, nothing; the kernel sits at the gather rate of the cache
// hierarchy, not at its occupancy.)
template <int VEC, bool P24, int SB>
__global__ __launch_bounds__(256) void k_score(const float* __restrict__ M_, const float* __restrict__ a_rank_,
                                               const int32_t* __restrict__ rb_off_, const int32_t* __restrict__ csr_idx_,
                                               const float* __restrict__ csr_e_, const float* __restrict__ csr_q_,
                                               const double* __restrict__ pvpi_,
                                               const int32_t* __restrict__ n_out_, float* __restrict__ S_, ScoreArgs A) {
    score_body<VEC, P24, SB>(M_, a_rank_, rb_off_, csr_idx_, csr_e_, csr_q_, pvpi_, n_out_, S_, A, (int)blockIdx.x);
}

// the seed pass of the one-cluster pruned job with the super-block bounds riding along (score_body<..., SUP = true>)
__global__ __launch_bounds__(256) void k_score_sup(const float* __restrict__ M_, const float* __restrict__ a_rank_,
                                                   const int32_t* __restrict__ rb_off_, const int32_t* __restrict__ csr_idx_,
                                                   const float* __restrict__ csr_e_, const float* __restrict__ csr_q_,
                                                   const double* __restrict__ pvpi_,
                                                   const int32_t* __restrict__ n_out_, float* __restrict__ S_, ScoreArgs A) {
    score_body<4, true, 8, true>(M_, a_rank_, rb_off_, csr_idx_, csr_e_, csr_q_, pvpi_, n_out_, S_, A, (int)blockIdx.x);
}

// ================================================================ top-N (PriorityQueue + poll loop, AbstractRM2Reducer.java:325, 358-369)
// One workgroup per user.  NaN marks "not a candidate" (rated by the user / padding).  Order: larger score first
// (IntDouble.compareTo, M/util/IntDouble.java:31-34); ties, unspecified in the reference, by ascending raw item id
// (at the cut-off of a truncated list: by ascending popularity rank).  Radix select on the order-preserving integer
// image of the float finds the K-th value in three passes over the row (L2 resident), a fourth pass collects.
struct TopNArgs {
    const float* __restrict__ S;
    int64_t ldS;
    int32_t Ic;
    const int32_t* __restrict__ n_out;     // by slot - slot_lo
    const int32_t* __restrict__ out_off;   // by slot - slot_lo (exclusive prefix of n_out)
    const int32_t* __restrict__ rank_item_raw;   // offset by pbase
    const int32_t* __restrict__ slot2du;
    const int32_t* __restrict__ uid;
    int32_t slot_lo, slot0, cluster;
    int32_t* __restrict__ out_user;
    int32_t* __restrict__ out_item;
    float* __restrict__ out_score;
    int32_t* __restrict__ out_cluster;
    // branch and bound (fy_rm2.hip, "exact pruning"): only the seed columns and the surviving 256-column blocks of a score row
    // are ever written.  mode 0: the whole row is live.  mode 1 (seed phase): sort the seed columns, publish tau_u and the
    // list the user gets if no block survives.  mode 2 (merge phase): users with surviving blocks only, seed + survivors.
    int32_t mode;
    int32_t seed_cols;
    const uint16_t* __restrict__ surv;    // [u * ldb + k]: surviving block ids
    const int32_t* __restrict__ n_quads;  // [u]: surviving blocks of the user
    int64_t ldb;
    float* __restrict__ tau;              // [u]
    // cooperative ranks: the survivors' scores are not in the (seed-wide) row but packed, 256 floats per surviving block,
    // at entry quad_prefix[u] + k; the user's blocks are in ascending order
    const float* __restrict__ Ssurv;
    const int32_t* __restrict__ quad_prefix;
};

__device__ __forceinline__ uint32_t fy_order_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fy_order_unkey(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

constexpr int TOPN_MAX = 2048;       // longest list (min(numberOfRecommendations, items of the cluster))
constexpr int TOPN_CAP = 4096;       // candidate buffer of k_topn_fast (LDS)
constexpr int TOPN_BINS = 4096;
constexpr int TOPN_SAMPLE = 1024;    // leading columns that give the lower bound of a short list; a long list takes 4 N (fy_topn_sample)
constexpr int TOPN_LONG = 256;       // lists longer than this take k_topn_long
constexpr int PRUNE_BLOCK_COLS = 256;   // candidate block of the branch and bound = one column chunk
constexpr int SEED_CHUNKS_MAX = 40;     // widest seed of the branch and bound: 5 N columns for the longest list (TOPN_MAX), in 256-column chunks

// descending bitonic sort of P2 (power of two) 64-bit keys in LDS; every thread of the block calls it
__device__ __forceinline__ void fy_bitonic_desc(uint64_t* v, int P2) {
    for (int k = 2; k <= P2; k <<= 1) {
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
}

// Fast path, one pass over the score row.  The columns are in popularity order and RM2 scores grow with the
// popularity of the candidate, so the K-th best of the first TOPN_SAMPLE columns is a tight LOWER bound tau of the
// true K-th best: sort the sample in LDS, keep its top K, then stream the rest of the row (16-byte loads) and keep
// only scores >= tau.  Every member of the true top K is among the kept ones (a member of the overall top K is in the
// top K of any subset that contains it), so an exact sort of the kept set gives the exact list, ties broken by
// ascending raw item id.  If more than TOPN_MAX entries survive (massive ties, e.g. a row of -inf) the user is
// flagged and k_topn_select below redoes it with a radix select.
__device__ __forceinline__ void topn_fast_body(const TopNArgs& A, int32_t* __restrict__ overflow, int32_t* __restrict__ any_overflow,
                                               int force_select, const int u /* user of the batch = work-group index */) {
    __shared__ uint64_t cand[TOPN_MAX];
    __shared__ uint32_t sh_count, sh_nvalid;
    const int slot = A.slot0 + u;
    const int K = A.n_out[slot - A.slot_lo];
    if (threadIdx.x == 0) overflow[u] = 0;
    if (A.mode == 2 && A.n_quads[u] == 0) return;         // the seed phase already wrote this user's final list
    if (K == 0) {
        if (A.mode == 1 && threadIdx.x == 0) A.tau[u] = INFINITY;   // nothing to emit: every block may be skipped
        return;
    }
    if (force_select && A.mode != 1) {   // test hook (FY_TOPN_FORCE_SELECT=1): exercise the radix-select path for every user
        if (threadIdx.x == 0) { overflow[u] = 1; atomicAdd(any_overflow, 1); }
        return;
    }
    const float* __restrict__ row = A.S + (int64_t)u * A.ldS;
    const int tid = threadIdx.x;
    constexpr int sample0 = TOPN_SAMPLE;
    const int Ls = min(A.Ic, A.mode ? min(A.seed_cols, TOPN_SAMPLE) : sample0);
    if (tid == 0) sh_nvalid = 0;
    __syncthreads();
    int myvalid = 0;
    int LP2 = 64;                      // sort only as much as the sample needs
    while (LP2 < Ls) LP2 <<= 1;
    for (int i = tid; i < LP2; i += blockDim.x) {
        uint64_t c = 0ull;
        if (i < Ls) {
            const float f = row[i];
            if (f == f) { c = ((uint64_t)fy_order_key(f) << 32) | (uint32_t)(0x7FFFFFFF - A.rank_item_raw[i]); myvalid++; }
        }
        cand[i] = c;
    }
    if (myvalid) atomicAdd(&sh_nvalid, (uint32_t)myvalid);
    __syncthreads();
    fy_bitonic_desc(cand, LP2);
    const int nvalid = (int)sh_nvalid;
    const uint32_t tau = nvalid >= K ? (uint32_t)(cand[K - 1] >> 32) : 0u;   // valid keys are > 0
    const int keep = min(K, nvalid);
    if (A.mode == 1) {
        // seed phase: tau for the bound pass, and the list that stands unless a block survives (almost always)
        if (tid == 0) A.tau[u] = nvalid >= K ? fy_order_unkey(tau) : -INFINITY;
        const int off1 = A.out_off[slot - A.slot_lo];
        const int user1 = A.uid[A.slot2du[slot]];
        for (int i = tid; i < keep; i += blockDim.x) {
            const uint64_t c = cand[i];
            A.out_user[off1 + i] = user1;
            A.out_item[off1 + i] = 0x7FFFFFFF - (int32_t)(uint32_t)c;
            A.out_score[off1 + i] = fy_order_unkey((uint32_t)(c >> 32));
            A.out_cluster[off1 + i] = A.cluster;
        }
        return;
    }
    __syncthreads();
    if (tid == 0) sh_count = (uint32_t)keep;
    __syncthreads();
    // stream the rest of the row (mode 2: only the surviving blocks, 16 float4 each)
    const int i4_begin = A.mode == 2 ? 0 : (sample0 >> 2);
    const int i4_end = A.mode == 2 ? A.n_quads[u] * (PRUNE_BLOCK_COLS / 4) : (A.Ic + 3) >> 2;
    auto take4 = [&](const float4& f4, int i4) __attribute__((always_inline)) {
        const float fv[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float f = fv[q];
            const int i = 4 * i4 + q;
            if (f == f && i < A.Ic) {
                const uint32_t key = fy_order_key(f);
                if (key >= tau) {
                    const uint32_t pos = atomicAdd(&sh_count, 1u);
                    if (pos < (uint32_t)TOPN_MAX) cand[pos] = ((uint64_t)key << 32) | (uint32_t)(0x7FFFFFFF - A.rank_item_raw[i]);
                }
            }
        }
    };
    if (A.mode == 2) {
        for (int x = i4_begin + tid; x < i4_end; x += blockDim.x) {
            const unsigned blk = A.surv[(int64_t)u * A.ldb + (x >> 6)];
            const int i4 = (int)blk * (PRUNE_BLOCK_COLS / 4) + (x & 63);
            const float* src = A.Ssurv ? A.Ssurv + ((int64_t)(A.quad_prefix[u] + (x >> 6)) * PRUNE_BLOCK_COLS + 4 * (x & 63)) : row + 4 * (int64_t)i4;
            take4(*reinterpret_cast<const float4*>(src), i4);
        }
    } else {
        // whole row: four 16-byte loads in flight per thread (a row is 236 KB at ML-25M shape; one load per trip was latency-bound)
        constexpr int UNR = 4;
        const float4 nan4 = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
        for (int x0 = i4_begin + tid; x0 < i4_end; x0 += UNR * (int)blockDim.x) {
            float4 f4[UNR];
#pragma unroll
            for (int q = 0; q < UNR; q++) {
                const int x = x0 + q * (int)blockDim.x;
                f4[q] = x < i4_end ? *reinterpret_cast<const float4*>(row + 4 * (int64_t)x) : nan4;
            }
#pragma unroll
            for (int q = 0; q < UNR; q++) take4(f4[q], x0 + q * (int)blockDim.x);
        }
    }
    __syncthreads();
    const int n = (int)sh_count;
    if (n > TOPN_MAX) {   // block-uniform
        if (tid == 0) { overflow[u] = 1; atomicAdd(any_overflow, 1); }
        return;
    }
    int P2 = 1;
    while (P2 < n) P2 <<= 1;
    for (int i = n + tid; i < P2; i += blockDim.x) cand[i] = 0ull;
    __syncthreads();
    if (n > keep) fy_bitonic_desc(cand, P2);     // nothing survived beyond the sorted sample head: already in order
    const int off = A.out_off[slot - A.slot_lo];
    const int user_raw = A.uid[A.slot2du[slot]];
    for (int i = tid; i < K; i += blockDim.x) {
        const uint64_t c = cand[i];
        A.out_user[off + i] = user_raw;
        A.out_item[off + i] = 0x7FFFFFFF - (int32_t)(uint32_t)c;
        A.out_score[off + i] = fy_order_unkey((uint32_t)(c >> 32));
        A.out_cluster[off + i] = A.cluster;
    }
}
__global__ __launch_bounds__(256) void k_topn_fast(TopNArgs A, int32_t* __restrict__ overflow, int32_t* __restrict__ any_overflow,
                                                   int force_select) {
    topn_fast_body(A, overflow, any_overflow, force_select, (int)blockIdx.x);
}

// Seed phase of the pruned flow (what k_topn_fast does in mode 1), one WAVE per user: the seed scores (256 columns for N = 50) are
// sorted in the wave's own slice of LDS with compiler fences instead of workgroup barriers -- a wave's LDS instructions complete in
// order -- then tau_u and the speculative list are published exactly as in mode 1.  (A 256-thread workgroup per user spent its
// time in the ~36 barriers of a 256-element bitonic sort: 1.3 ms per job for 162 541 users.)
__device__ __forceinline__ void fy_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Round 4: the sort lives in REGISTERS -- element i = 64 r + lane is lane's register r; a compare-exchange over a distance of 64 or more
// is between two registers of one lane, a shorter one between two lanes (two 32-bit cross-lane moves per element), no LDS memory at
// all.  (The LDS version read and wrote eight 64-bit words per lane and stage behind a fence: 0.74 ms per job for 162 541 users.)
template <int R>
__global__ __launch_bounds__(256) void k_topn_seed(TopNArgs A, int32_t n_users, int32_t* __restrict__ overflow) {
    constexpr int LP2 = 64 * R;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int u = blockIdx.x * 4 + w;
    if (u >= n_users) return;                        // wave-uniform; nothing below synchronises the workgroup
    const int slot = A.slot0 + u;
    const int K = A.n_out[slot - A.slot_lo];
    if (lane == 0) overflow[u] = 0;
    if (K == 0) {
        if (lane == 0) A.tau[u] = INFINITY;          // nothing to emit: every block may be skipped
        return;
    }
    const float* __restrict__ row = A.S + (int64_t)u * A.ldS;
    const int Ls = min(A.Ic, min(A.seed_cols, TOPN_SAMPLE));
    unsigned long long v[R];
    int myvalid = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = 64 * r + lane;
        unsigned long long c = 0ull;
        if (i < Ls) {
            const float f = row[i];
            if (f == f) { c = ((unsigned long long)fy_order_key(f) << 32) | (uint32_t)(0x7FFFFFFF - A.rank_item_raw[i]); myvalid++; }
        }
        v[r] = c;
    }
    for (int o = 32; o > 0; o >>= 1) myvalid += __shfl_xor(myvalid, o, 64);
    // bitonic sort, descending, of the LP2 keys (invalid ones are 0: they end up last)
#pragma unroll
    for (int k = 2; k <= LP2; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int jr = j >> 6;
                    if ((r & jr) == 0) {
                        const int r2 = r | jr;
                        const bool desc = ((64 * r) & k) == 0;      // (k >= 128 here: bit k of i = 64 r + lane is a bit of r)
                        const unsigned long long x = v[r], y = v[r2];
                        const bool sw = desc ? (x < y) : (x > y);
                        v[r] = sw ? y : x;
                        v[r2] = sw ? x : y;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int i = 64 * r + lane;
                    const unsigned long long x = v[r];
                    const unsigned long long y = __shfl_xor(x, j, 64);
                    const bool lower = (lane & j) == 0, desc = (i & k) == 0;
                    const bool keep_max = lower == desc;
                    v[r] = keep_max ? (x > y ? x : y) : (x < y ? x : y);
                }
            }
        }
    }
    const int nvalid = myvalid;
    const int keep = min(K, nvalid);
    {   // tau_u = the K-th best seed score: element K - 1
        unsigned long long t = v[0];
#pragma unroll
        for (int r = 1; r < R; r++) t = ((K - 1) >> 6) == r ? v[r] : t;
        t = __shfl(t, (K - 1) & 63, 64);
        if (lane == 0) A.tau[u] = nvalid >= K ? fy_order_unkey((uint32_t)(t >> 32)) : -INFINITY;
    }
    const int off1 = A.out_off[slot - A.slot_lo];
    const int user1 = A.uid[A.slot2du[slot]];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = 64 * r + lane;
        if (i < keep) {
            const unsigned long long c = v[r];
            A.out_user[off1 + i] = user1;
            A.out_item[off1 + i] = 0x7FFFFFFF - (int32_t)(uint32_t)c;
            A.out_score[off1 + i] = fy_order_unkey((uint32_t)(c >> 32));
            A.out_cluster[off1 + i] = A.cluster;
        }
    }
}

// Long lists (N > TOPN_LONG; the reference's default is N = 1000, RMRecommenderDriver.java:95), whole rows.  With a 1024-column
// sample the 1000th best is no bound at all: in round 2 every user overflowed into the multi-pass radix select below
// (117 ms per job for 38 GB of scores, 213 ms at 50 clusters).  Here the sample is the 4 N most popular columns (at most 4096),
// and instead of sorting it the threshold comes from a two-level histogram of the sample's keys (bits 31..24, then 23..16): the
// largest 16-bit key prefix T with at least N sample values at or above it -- a valid lower bound of the row's N-th best, 2^-7
// relative below the sample's own N-th best, so the sample contributes N + a few dozen candidates.  One stream over the rest of
// the row appends what reaches T; one bitonic sort of the ~N + 100 candidates gives the list (ties by ascending raw item id).
// More than TOPN_CAP candidates (massive ties): the user is flagged for k_topn_select, like in k_topn_fast.
__device__ __forceinline__ void fy_topn_bin(const uint32_t* __restrict__ hist, uint32_t above, uint32_t K, uint32_t* __restrict__ out /* [0] bin, [1] count above it */) {
    // wave 0: lane l owns bins 4 l .. 4 l + 3; the highest bin b with above + (count in bins >= b) >= K (bin 0 if none)
    const int lane = threadIdx.x & 63;
    const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    uint32_t suf = h0 + h1 + h2 + h3;                       // -> inclusive suffix sum over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_down((int)suf, o, 64);
        if (lane + o < 64) suf += t;
    }
    const uint32_t higher = above + suf - (h0 + h1 + h2 + h3);     // everything in the lanes above this one
    const unsigned long long ok = __ballot(higher + h0 + h1 + h2 + h3 >= K);
    if (!ok) { if (lane == 0) { out[0] = 0u; out[1] = above + suf - h0; } return; }      // fewer than K in all: bin 0 takes everything
    const int L = 63 - __clzll((long long)ok);
    if (lane == L) {
        uint32_t cum = higher;
        int b = 3;
        const uint32_t hh[4] = {h0, h1, h2, h3};
        for (; b > 0; b--) {
            if (cum + hh[b] >= K) break;
            cum += hh[b];
        }
        out[0] = (uint32_t)(4 * L + b);
        out[1] = cum;
    }
}
// candidate buffer (dynamic LDS): N + a few dozen entries are expected -- 2048 (16 KB, eight workgroups per CU) up to N = 1400, else 4096
inline int fy_topn_long_cap(int top_n) { return top_n <= 1400 ? 2048 : TOPN_CAP; }
__global__ __launch_bounds__(256) void k_topn_long(TopNArgs A, int32_t* __restrict__ overflow, int32_t* __restrict__ any_overflow, int force_select, int cap) {
    // A.mode (round 4: the branch and bound for long lists, seeds of up to SEED_CHUNKS_MAX * 256 columns):
    //   0  whole rows (the plain full pass);
    //   1  seed phase of the pruned flow: the row holds the seed columns only; publishes tau_u = the K-th best seed score and the list
    //      that stands unless a block survives.  A user whose candidates overflow the buffer (massive ties) gets the histogram
    //      threshold as tau -- a valid lower bound of its K-th best -- and overflow[u] = 1, which the merge launch below reads;
    //   2  merge phase: users with surviving blocks (or a seed overflow): the seed row and the packed survivor scores against the
    //      exact tau_u; anything that does not fit goes to k_topn_select.
    extern __shared__ __attribute__((aligned(16))) uint64_t fy_topn_cand[];
    uint64_t* __restrict__ cand = fy_topn_cand;
    __shared__ uint32_t hist[256], sh_bin[2], sh_count;
    const int u = blockIdx.x, tid = threadIdx.x;
    const int slot = A.slot0 + u;
    const int K = A.n_out[slot - A.slot_lo];
    const int mode = A.mode;
    const int seed_ovf = mode == 2 ? overflow[u] : 0;       // (written by this batch's mode-1 launch, in stream order)
    __syncthreads();
    if (tid == 0) overflow[u] = 0;
    if (mode == 2 && A.n_quads[u] == 0 && !seed_ovf) return;     // the seed phase already wrote this user's final list
    if (K == 0) {
        if (mode == 1 && tid == 0) A.tau[u] = INFINITY;          // nothing to emit: every block may be skipped
        return;
    }
    if ((force_select && mode != 1) || seed_ovf) {
        if (tid == 0) { overflow[u] = 1; atomicAdd(any_overflow, 1); }
        return;
    }
    const float* __restrict__ row = A.S + (int64_t)u * A.ldS;
    const int row_cols = mode ? min(A.Ic, A.seed_cols) : A.Ic;   // pruned flow: only the seed columns of a row exist
    int sample = TOPN_SAMPLE;
    while (sample < 4 * K && sample < TOPN_CAP) sample <<= 1;
    const int Ls = min(row_cols, sample);
    constexpr int PER = TOPN_CAP / 256;
    uint32_t key[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int i = tid + 256 * q;
        const float f = i < Ls ? row[i] : __builtin_nanf("");
        key[q] = f == f ? fy_order_key(f) : 0u;              // valid keys are > 0
    }
    uint32_t tau;
    if (mode == 2) {
        tau = fy_order_key(A.tau[u]);                            // exact K-th best of the seed (-inf: fewer than K seed scores)
    } else {
        uint32_t prefix = 0, above = 0;
        for (int level = 0; level < 2; level++) {
            const int shift = level == 0 ? 24 : 16;
            hist[tid] = 0u;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PER; q++)
                if (key[q] && (level == 0 || (key[q] >> 24) == (prefix >> 24))) atomicAdd(&hist[(key[q] >> shift) & 255u], 1u);
            __syncthreads();
            if (tid < 64) fy_topn_bin(hist, above, (uint32_t)K, sh_bin);
            __syncthreads();
            prefix |= sh_bin[0] << shift;
            above = sh_bin[1];
            __syncthreads();
        }
        tau = prefix ? prefix : 1u;                              // fewer than K valid sample values: every valid score is a candidate
    }
    if (tid == 0) sh_count = 0u;
    __syncthreads();
    auto take = [&](uint32_t k, int i) __attribute__((always_inline)) {
        if (k >= tau) {                                        // (invalid: key 0 < tau)
            const uint32_t pos = atomicAdd(&sh_count, 1u);
            if (pos < (uint32_t)cap) cand[pos] = ((uint64_t)k << 32) | (uint32_t)(0x7FFFFFFF - A.rank_item_raw[i]);
        }
    };
#pragma unroll
    for (int q = 0; q < PER; q++) take(key[q], tid + 256 * q);
    constexpr int UNR = 4;
    const float4 nan4 = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
    const int i4_end = (row_cols + 3) >> 2;
    for (int x0 = (sample >> 2) + tid; x0 < i4_end; x0 += UNR * 256) {
        float4 f4[UNR];
#pragma unroll
        for (int q = 0; q < UNR; q++) {
            const int x = x0 + q * 256;
            f4[q] = x < i4_end ? *reinterpret_cast<const float4*>(row + 4 * (int64_t)x) : nan4;
        }
#pragma unroll
        for (int q = 0; q < UNR; q++) {
            const float fv[4] = {f4[q].x, f4[q].y, f4[q].z, f4[q].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = 4 * (x0 + q * 256) + e;
                const float f = fv[e];
                take((f == f && i < row_cols) ? fy_order_key(f) : 0u, i);
            }
        }
    }
    if (mode == 2) {      // the surviving blocks: 64 float4 each, packed at quad_prefix[u] + k (or in place, cooperative ranks aside)
        const int n4 = A.n_quads[u] * (PRUNE_BLOCK_COLS / 4);
        for (int x = tid; x < n4; x += 256) {
            const unsigned blk = A.surv[(int64_t)u * A.ldb + (x >> 6)];
            const int i4 = (int)blk * (PRUNE_BLOCK_COLS / 4) + (x & 63);
            const float* src = A.Ssurv ? A.Ssurv + ((int64_t)(A.quad_prefix[u] + (x >> 6)) * PRUNE_BLOCK_COLS + 4 * (x & 63)) : row + 4 * (int64_t)i4;
            const float4 f4 = *reinterpret_cast<const float4*>(src);
            const float fv[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = 4 * i4 + e;
                const float f = fv[e];
                take((f == f && i < A.Ic) ? fy_order_key(f) : 0u, i);
            }
        }
    }
    __syncthreads();
    const int n = (int)sh_count;
    if (mode == 1) {
        if (n > cap) {       // block-uniform: no list yet; the histogram threshold is a lower bound of the K-th best all the same
            if (tid == 0) { A.tau[u] = fy_order_unkey(tau); overflow[u] = 1; }
            return;
        }
    } else if (n > cap || n < K) {   // block-uniform
        // (n < K cannot happen while n_out counts only scored candidates and the histogram threshold keeps >= K of the sample -- but if it
        // ever did, e.g. NaN scores, the entries behind n would be emitted as items: such a user goes to the exact select instead)
        if (tid == 0) { overflow[u] = 1; atomicAdd(any_overflow, 1); }
        return;
    }
    int P2 = 1;
    while (P2 < n) P2 <<= 1;
    for (int i = n + tid; i < P2; i += 256) cand[i] = 0ull;
    __syncthreads();
    fy_bitonic_desc(cand, P2);
    if (mode == 1 && tid == 0) A.tau[u] = n >= K ? fy_order_unkey((uint32_t)(cand[K - 1] >> 32)) : -INFINITY;
    const int off = A.out_off[slot - A.slot_lo];
    const int user_raw = A.uid[A.slot2du[slot]];
    for (int i = tid; i < min(K, n); i += 256) {
        const uint64_t c = cand[i];
        A.out_user[off + i] = user_raw;
        A.out_item[off + i] = 0x7FFFFFFF - (int32_t)(uint32_t)c;
        A.out_score[off + i] = fy_order_unkey((uint32_t)(c >> 32));
        A.out_cluster[off + i] = A.cluster;
    }
}

// Fallback: exact radix select (three histogram passes + collect).  Runs only for users k_topn_fast flagged.
__device__ __forceinline__ void topn_select_body(const TopNArgs& A, const int32_t* __restrict__ overflow,
                                                 const int32_t* __restrict__ any_overflow, unsigned long long* __restrict__ n_selected, const int u) {
    __shared__ uint32_t hist[TOPN_BINS];
    __shared__ uint64_t cand[TOPN_MAX];
    __shared__ uint32_t sh_prefix, sh_need, sh_count, sh_eq_taken;
    if (*any_overflow == 0) return;
    if (overflow[u] == 0) return;
    const int slot = A.slot0 + u;
    const int K = A.n_out[slot - A.slot_lo];
    if (K == 0) return;
    const float* __restrict__ row = A.S + (int64_t)u * A.ldS;
    const int tid = threadIdx.x;
    if (tid == 0 && n_selected) atomicAdd(n_selected, 1ull);
    // pruned rows: only the seed columns and the surviving 256-column blocks were ever written
    __shared__ uint32_t live[2048];
    __shared__ uint16_t live_pre[2048];   // cooperative ranks: surviving blocks in front of word w (position in the packed scores)
    if (A.mode) {
        for (int w = tid; w < 2048; w += blockDim.x) live[w] = 0;
        __syncthreads();
        for (int k = tid; k < A.n_quads[u]; k += blockDim.x) {
            const unsigned blk = A.surv[(int64_t)u * A.ldb + k];
            atomicOr(&live[blk >> 5], 1u << (blk & 31u));
        }
        __syncthreads();
        if (A.Ssurv && tid == 0) {
            unsigned run = 0;
            for (int w = 0; w < 2048; w++) { live_pre[w] = (uint16_t)run; run += __popc(live[w]); }
        }
        __syncthreads();
    }
    const float* __restrict__ packed = (A.mode && A.Ssurv) ? A.Ssurv + (int64_t)A.quad_prefix[u] * PRUNE_BLOCK_COLS : nullptr;
    auto rowval = [&](int i) -> float {
        if (A.mode == 0 || i < A.seed_cols) return row[i];
        const unsigned blk = (unsigned)i >> 8, w = blk >> 5, bit = blk & 31u;
        if (!((live[w] >> bit) & 1u)) return __builtin_nanf("");
        if (!packed) return row[i];
        const unsigned pos = live_pre[w] + __popc(live[w] & ((1u << bit) - 1u));
        return packed[(int64_t)pos * PRUNE_BLOCK_COLS + (i & 255)];
    };
#define FY_ROWVAL(i) rowval(i)

    // ---- radix select: 12 + 10 + 10 bits, most significant first
    uint32_t prefix = 0, prefix_mask = 0, need = (uint32_t)K;
    const int shifts[3] = {20, 10, 0};
    const int widths[3] = {12, 10, 10};
    for (int pass = 0; pass < 3; pass++) {
        const int nb = 1 << widths[pass];
        for (int b = tid; b < nb; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        for (int i = tid; i < A.Ic; i += blockDim.x) {
            const float f = FY_ROWVAL(i);
            if (f != f) continue;
            const uint32_t key = fy_order_key(f);
            if ((key & prefix_mask) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t cum = 0;
            int b = nb - 1;
            for (; b > 0; b--) {
                if (cum + hist[b] >= need) break;
                cum += hist[b];
            }
            sh_prefix = prefix | ((uint32_t)b << shifts[pass]);
            sh_need = need - cum;
        }
        __syncthreads();
        prefix = sh_prefix;
        need = sh_need;
        prefix_mask |= (uint32_t)(nb - 1) << shifts[pass];
        __syncthreads();
    }
    const uint32_t T = prefix;      // key of the K-th largest candidate; `need` of the entries equal to T are taken
    // after the last pass hist[] counts exact keys: how many candidates tie with the K-th value
    const uint32_t eq_total = hist[T & 1023u];
    const bool take_all_eq = (eq_total == need);   // block-uniform; the common case (no tie across the cut-off)
    __syncthreads();

    // ---- collect everything above T (and the ties when all of them fit); order is fixed by the sort below
    if (tid == 0) { sh_count = 0; sh_eq_taken = 0; }
    __syncthreads();
    for (int i = tid; i < A.Ic; i += blockDim.x) {
        const float f = FY_ROWVAL(i);
        if (f != f) continue;
        const uint32_t key = fy_order_key(f);
        if (key > T || (take_all_eq && key == T)) {
            const uint32_t pos = atomicAdd(&sh_count, 1u);
            if (pos < (uint32_t)TOPN_MAX) cand[pos] = ((uint64_t)key << 32) | (uint32_t)(0x7FFFFFFF - A.rank_item_raw[i]);
        }
    }
    __syncthreads();
    if (!take_all_eq) {
        // a tie straddles the cut-off: take the first `need` tied entries in index (popularity-rank) order.
        // Block-uniform loop: every thread reaches every barrier.
        uint32_t* wave_cnt = hist;   // reuse
        const int w = tid >> 6, ln = tid & 63, nw = blockDim.x >> 6;
        for (int base = 0; base < A.Ic; base += blockDim.x) {
            const int i = base + tid;
            bool eq = false;
            if (i < A.Ic) {
                const float f = FY_ROWVAL(i);
                eq = (f == f) && fy_order_key(f) == T;
            }
            const unsigned long long bal = __ballot(eq);
            if (ln == 0) wave_cnt[w] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t off = sh_eq_taken, all = 0;
            for (int x = 0; x < nw; x++) {
                if (x < w) off += wave_cnt[x];
                all += wave_cnt[x];
            }
            off += (uint32_t)__popcll(bal & ((1ull << ln) - 1ull));
            if (eq && off < need) {
                const uint32_t pos = atomicAdd(&sh_count, 1u);
                if (pos < (uint32_t)TOPN_MAX) cand[pos] = ((uint64_t)T << 32) | (uint32_t)(0x7FFFFFFF - A.rank_item_raw[i]);
            }
            __syncthreads();
            if (tid == 0) sh_eq_taken += all;
            __syncthreads();
            if (sh_eq_taken >= need) break;   // uniform: read after the barrier
        }
        __syncthreads();
    }
    const int n = min((int)sh_count, K);     // == K
    int P2 = 1;
    while (P2 < n) P2 <<= 1;
    for (int i = n + tid; i < P2; i += blockDim.x) cand[i] = 0ull;
    __syncthreads();
    // ---- bitonic sort, descending on (score key, ~item)
    for (int k = 2; k <= P2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P2; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const uint64_t x = cand[i], y = cand[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { cand[i] = y; cand[l] = x; }
                }
            }
            __syncthreads();
        }
    }
    const int off = A.out_off[slot - A.slot_lo];
    const int user_raw = A.uid[A.slot2du[slot]];
    for (int i = tid; i < n; i += blockDim.x) {
        const uint64_t c = cand[i];
        A.out_user[off + i] = user_raw;
        A.out_item[off + i] = 0x7FFFFFFF - (int32_t)(uint32_t)c;
        A.out_score[off + i] = fy_order_unkey((uint32_t)(c >> 32));   // already the (float) cast of RM2HDFSReducer.java:48
        A.out_cluster[off + i] = A.cluster;
    }
}
__global__ __launch_bounds__(256) void k_topn_select(TopNArgs A, const int32_t* __restrict__ overflow,
                                                     const int32_t* __restrict__ any_overflow, unsigned long long* __restrict__ n_selected = nullptr) {
    topn_select_body(A, overflow, any_overflow, n_selected, (int)blockIdx.x);
}

// ================================================================ many small clusters in ONE launch per kernel ("flat batch")
// A job over many clusters that are too small for the branch and bound (ML-1M shape in 50 clusters: 120 users and ~2 000 items
// each) was a chain of ~9 stream operations per cluster -- 450 launches whose kernels last 60-120 us and cannot fill the chip;
// 10 ms per job, most of it launch latency.  Here every kernel of the chain runs ONCE for all those clusters: blockIdx.y = the
// cluster, its arguments come from a descriptor array in device memory (wave-uniform: scalar loads), work-groups beyond the
// cluster's own grid leave at once.  The bodies are the single-cluster kernels' (score_body, topn_fast_body, topn_select_body).
struct FlatDesc {
    ScoreArgs SA;
    TopNArgs TA;
    int32_t* overflow;            // [n_users]
    int32_t* any_overflow;        // this cluster's flag
    int32_t* n_heavy;             // this cluster's count (= SA.n_heavy)
    int32_t score_grid, n_users, heavy_thresh, pad;
};
__global__ void k_count_heavy_multi(const FlatDesc* __restrict__ D, int32_t n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FlatDesc& d = D[i];
    const int32_t* __restrict__ rowptr = d.SA.rb_off + (d.SA.slot0 - d.SA.slot_base);
    int lo = 0, hi = d.n_users;       // first u with degree <= thresh (slots are in descending order of degree)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (rowptr[mid + 1] - rowptr[mid] > d.heavy_thresh) lo = mid + 1; else hi = mid; }
    *d.n_heavy = lo;
    *d.any_overflow = 0;
}
template <int VEC, bool P24, int SB>
__global__ __launch_bounds__(256) void k_score_multi(const FlatDesc* __restrict__ D, const int32_t* __restrict__ csr_idx_,
                                                     const float* __restrict__ csr_e_, const float* __restrict__ csr_q_,
                                                     const double* __restrict__ pvpi_, const int32_t* __restrict__ n_out_) {
    const FlatDesc& d = D[blockIdx.y];
    if ((int)blockIdx.x >= d.score_grid) return;
    const ScoreArgs A = d.SA;
    score_body<VEC, P24, SB>(A.M, A.a_rank, A.rb_off, csr_idx_, csr_e_, csr_q_, pvpi_, n_out_, A.S, A, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void k_topn_fast_multi(const FlatDesc* __restrict__ D, int force_select) {
    const FlatDesc& d = D[blockIdx.y];
    if ((int)blockIdx.x >= d.n_users) return;
    const TopNArgs A = d.TA;
    topn_fast_body(A, d.overflow, d.any_overflow, force_select, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void k_topn_select_multi(const FlatDesc* __restrict__ D, unsigned long long* __restrict__ n_selected) {
    const FlatDesc& d = D[blockIdx.y];
    if ((int)blockIdx.x >= d.n_users) return;
    const TopNArgs A = d.TA;
    topn_select_body(A, d.overflow, d.any_overflow, n_selected, (int)blockIdx.x);
}

// ================================================================ exact pruning of candidate blocks (branch and bound)
// For a block B of 256 candidate columns,
//     UB(u, B) = pvpi + sum_{j in rated(u)} ln( max_{i in B} G[j][i] + q_j max_{i in B} b_i + (max_{i in B} a_i) * e_uj )  >=  score(u, i)
// for all i in B, because every term is monotone in G[j][i], b_i and a_i (all factors are non-negative).  Evaluating UB is the scoring kernel itself run on the reduced
// matrix Bmax[j][B] (1/256 of the columns).  With tau_u = the N-th best EXACT score among the first `seed` (most
// popular) columns, a block whose UB is below tau_u cannot contribute to the user's top N and is skipped; the exact
// kernel then runs only on the surviving (user, block) pairs.  RM2 scores fall steeply with candidate popularity, so on
// MovieLens-shaped data well under 1 % of the tail blocks survive -- the lists are bit-for-bit those of the full pass.
constexpr int PRUNE_BLOCK = 256;   // = the column chunk of the scoring kernel: a surviving block is one (user, chunk) work item

// block maxima of a = lambda * p and of b (the rank-one part of a term, q_j b_i + a_i e_uj, is monotone in both)
__global__ void k_block_amax(int32_t Ic, int32_t ldb, const float* __restrict__ a_rank, const float* __restrict__ b_rank,
                             float* __restrict__ amax, float* __restrict__ bmax, int32_t width = PRUNE_BLOCK) {
    for (int32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < ldb; b += gridDim.x * blockDim.x) {
        float m = 0.0f, mb = 0.0f;
        for (int64_t i = (int64_t)b * width; i < min((int64_t)Ic, (int64_t)(b + 1) * width); i++) { m = fmaxf(m, a_rank[i]); mb = fmaxf(mb, b_rank[i]); }
        amax[b] = m;
        bmax[b] = mb;
    }
}

// One WORKGROUP = one (user, surviving block): its four waves split the user's rated items into quarters and their partial
// log-sums are added in wave order (fixed, so the result is reproducible).  Survivors belong to the heaviest users -- a
// single wave walked thousands of rows in dependent batches of 8 and the pass ended on a handful of such waves.
template <int SB>
__global__ __launch_bounds__(256) void k_score_blocks(const float* __restrict__ M_, const float* __restrict__ a_rank_,
                                                      const int32_t* __restrict__ rowptr_, const int32_t* __restrict__ csr_idx_,
                                                      const float* __restrict__ csr_e_, const float* __restrict__ csr_q_,
                                                      const double* __restrict__ pvpi_,
                                                      const int32_t* __restrict__ surv_prefix_, const uint16_t* __restrict__ surv_,
                                                      float* __restrict__ S_, ScoreArgs A, int64_t ldb,
                                                      unsigned long long* __restrict__ counters, int32_t panel_blocks = 0x7FFFFFFF,
                                                      const uint8_t* __restrict__ surv_mask_ = nullptr) {
    __shared__ double sh_t[3][64][4];
    __shared__ unsigned sh_mask[3][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int total = surv_prefix_[A.n_users];
    const int64_t pitch = A.ldm * 3;
    const double LN2 = 0.69314718055994530942;
    const float qnan = __builtin_nanf("");
    unsigned long long my_terms = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0 && total) atomicAdd(&counters[0], (unsigned long long)total);
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        int lo = 0, hi = A.n_users;                 // last user with surv_prefix <= w
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (surv_prefix_[mid] <= w) lo = mid; else hi = mid;
        }
        const int u = lo;
        const int slot = A.slot0 + u;
        const int blk = (int)surv_[(int64_t)u * ldb + (w - surv_prefix_[u])];
        if (blk >= panel_blocks) continue;             // a block behind the column panel: k_score_stray's (block-uniform)
        // panel mode: only the 64-column sub-blocks that passed the bounds are read and scored (16 lanes each); the others' lanes
        // load nothing -- a surviving block has 1.3 live sub-blocks on average, and the pass is bound by the row segments it reads
        const bool live = !surv_mask_ || ((surv_mask_[(int64_t)u * ldb + (w - surv_prefix_[u])] >> (lane >> 4)) & 1u);
        const int col0 = blk * PRUNE_BLOCK;
        const int col = col0 + lane * 4;
        float a[4], bb[4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            a[v] = col + v < A.Ic ? a_rank_[col + v] : 0.0f;
            bb[v] = col + v < A.Ic ? A.b_rank[col + v] : 0.0f;
        }
        const char* __restrict__ Mcol = reinterpret_cast<const char*>(M_) + (int64_t)col * 3;
        const int row_beg = rowptr_[slot], row_end = rowptr_[slot + 1];
        const int quarter = (((row_end - row_beg + 3) >> 2) + SB - 1) / SB * SB;   // whole batches per wave
        const int beg = min(row_end, row_beg + wave * quarter), end = min(row_end, beg + quarter);
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        unsigned mask = 0;
        // (idx, e, q) of a batch by three vector loads, one batch ahead, handed out with v_readlane: see k_score's walk
        const int sub = lane & (SB - 1);
        int vi = 0, vi_n;
        float ve = 0.f, vq = 0.f, ve_n, vq_n;
        if (beg < end) {
            const int kk = min(beg + sub, end - 1);
            vi = csr_idx_[kk]; ve = csr_e_[kk]; vq = csr_q_[kk];
        }
        for (int k = beg; k < end; k += SB) {
            U3 g[SB];
            float e[SB], qq[SB];
            int jj[SB];
#pragma unroll
            for (int x = 0; x < SB; x++) {
                jj[x] = __builtin_amdgcn_readlane(vi, x);
                e[x] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ve), x));
                qq[x] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vq), x));
                g[x] = U3{0u, 0u, 0u};
                if (live) g[x] = *reinterpret_cast<const U3*>(Mcol + (int64_t)jj[x] * pitch);
            }
            {
                const int kk = min(k + SB + sub, end - 1);
                vi_n = csr_idx_[kk]; ve_n = csr_e_[kk]; vq_n = csr_q_[kk];
            }
            float p[4] = {1.f, 1.f, 1.f, 1.f};     // (product of the mantissas in fp32, exponents exact: see k_score's walk)
            int pe[4] = {0, 0, 0, 0};
#pragma unroll
            for (int x = 0; x < SB; x++) {
                if (k + x < end) {
                    float gv[4];
                    fy_unpack24(g[x], gv);
#pragma unroll
                    for (int v = 0; v < 4; v++) fy_logprod_step(fmaf(qq[x], bb[v], fmaf(a[v], e[x], gv[v])), p[v], pe[v]);
                    const unsigned d = (unsigned)(jj[x] - col);
                    if (d < 4u) mask |= 1u << d;
                }
            }
#pragma unroll
            for (int v = 0; v < 4; v++) t[v] += fy_logprod_fold(p[v], pe[v]);
            vi = vi_n; ve = ve_n; vq = vq_n;
        }
        if (wave > 0) {
#pragma unroll
            for (int v = 0; v < 4; v++) sh_t[wave - 1][lane][v] = t[v];
            sh_mask[wave - 1][lane] = mask;
        }
        __syncthreads();
        if (wave == 0) {
            for (int x = 0; x < 3; x++) {
#pragma unroll
                for (int v = 0; v < 4; v++) t[v] += sh_t[x][lane][v];
                mask |= sh_mask[x][lane];
            }
            const double base = pvpi_[slot - A.slot_lo];
            float4 o;
            o.x = (!live || (mask & 1u) || col + 0 >= A.Ic) ? qnan : (float)(base + LN2 * t[0]);
            o.y = (!live || (mask & 2u) || col + 1 >= A.Ic) ? qnan : (float)(base + LN2 * t[1]);
            o.z = (!live || (mask & 4u) || col + 2 >= A.Ic) ? qnan : (float)(base + LN2 * t[2]);
            o.w = (!live || (mask & 8u) || col + 3 >= A.Ic) ? qnan : (float)(base + LN2 * t[3]);
            // packed: entry w holds the block's 256 scores (the top-N kernels find them through quad_prefix)
            *reinterpret_cast<float4*>(S_ + (int64_t)w * PRUNE_BLOCK + lane * 4) = o;
            my_terms += (unsigned long long)(row_end - row_beg) * 256ull;
        }
        __syncthreads();   // sh_t is free again
    }
    if (wave == 0 && lane == 0 && my_terms) atomicAdd(&counters[1], my_terms);
}

// ================================================================ column-panel mode (many clusters)
// With many clusters a dense I_c x I_c matrix per cluster is what the job spends its time on (50 x 7.8 GB written and re-read at
// ML-25M shape), although only the seed columns, the block bounds and the few blocks that survive the bound are ever read.
// Panel mode keeps per cluster: Gp = the first `panel_cols` columns of every row (the seed columns and the popular blocks --
// where, measured, the surviving blocks are), and the block maxima at 64-COLUMN granularity for the whole row (in a sparse
// cluster the maximum of a 256-column block is almost always a co-rated pair while most of its candidates are not co-rated
// with most of the user's items: 64-column bounds cut the survivors ~4x).  A 256-column block survives when one of its four
// sub-blocks does; survivors inside the panel are scored from Gp by k_score_blocks, the rare survivors behind it ("strays":
// tail blocks of users with a dozen ratings) exactly from the sparse data by k_score_stray.
__global__ __launch_bounds__(256) void k_bound_select_sub(const float* __restrict__ UB64, int64_t ldb64, int32_t nsub, int32_t nblk, int32_t seed_blocks,
                                                          const float* __restrict__ tau, const double* __restrict__ pvpi /* [u] */,
                                                          int32_t n_users, int64_t ldsurv, uint16_t* __restrict__ surv, uint8_t* __restrict__ surv_mask,
                                                          int32_t* __restrict__ n_surv) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int u = blockIdx.x * wpb + (threadIdx.x >> 6); u < n_users; u += gridDim.x * wpb) {
        const float t = tau[u];
        const float pv = (float)pvpi[u];
        int count = 0;
        for (int b0 = 0; b0 < nblk; b0 += 64) {
            const int b = b0 + lane;
            unsigned keep = 0;     // bit q: sub-block q of the block survives
            if (b < nblk && b >= seed_blocks) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int sb = 4 * b + q;
                    if (sb < nsub && fy_bound_keeps(UB64[(int64_t)u * ldb64 + sb], t, pv)) keep |= 1u << q;
                }
            }
            const unsigned long long bal = __ballot(keep != 0);
            if (keep) {
                const int64_t at = (int64_t)u * ldsurv + count + __popcll(bal & ((1ull << lane) - 1ull));
                surv[at] = (uint16_t)b;
                surv_mask[at] = (uint8_t)keep;
            }
            count += __popcll(bal);
        }
        if (lane == 0) n_surv[u] = count;
    }
}

// Bound repair (panel mode).  The bound of user u for a 64-column sub-block takes, for every rated item j, the LARGEST G[j][i] of
// the sub-block -- also when that i is an item u rated itself, which is no candidate.  For a user with a few dozen ratings this
// is the rule, not the exception: G[j][i0] of a rated tail item i0 carries u's own x_ui0 x_uj (large: x = r / s_u) for every j
// of the list, so i0's sub-block is bounded high in every term while its 63 real candidates score ~100 below the threshold.
// Measured (ML-25M shape, 50 clusters): ALL surviving sub-blocks behind the panel and half of those inside it are of this kind.
// For the surviving sub-blocks that hold an item of the user's list the bound is therefore evaluated again with
//     exact rows (Bmax64 from the row kernel):  the second largest value of the sub-block when the column of the largest is
//                                               rated by u (Brep) -- an upper bound of every other column;
//     tail rows behind p_eff (k_tail_blocks):   the stored sum over raters minus u's own term x_uj * max_{i in sub-block} x_ui
//                                               (u contributes nothing to a column it did not rate),
// and the sub-block is dropped when the new bound stays below the threshold; blocks left without sub-blocks are removed from
// the user's list.  One wave per user; users with more than REPAIR_MAX_LIST ratings are left alone (their own terms are small).
constexpr int REPAIR_MAX_LIST = 256;
struct RepairArgs {
    uint16_t* __restrict__ surv;
    uint8_t* __restrict__ surv_mask;
    int32_t* __restrict__ n_surv;
    int64_t ldsurv;
    int32_t n_users, slot0, slot_lo, p_eff, Ic;
    const int32_t* __restrict__ rowptr;
    const int32_t* __restrict__ csr_idx;
    const float* __restrict__ csr_x;
    const float* __restrict__ csr_e;
    const float* __restrict__ csr_q;
    const float* __restrict__ Bmax64;      // 3 bytes per entry
    const uint32_t* __restrict__ Brep;
    int64_t ldb64;
    const float* __restrict__ amax64;
    const float* __restrict__ bmax64;
    const float* __restrict__ tau;         // [u]
    const double* __restrict__ pvpi;       // [slot - slot_lo]
    float w2;
    unsigned long long* __restrict__ counters;   // [4] sub-blocks dropped
};
__device__ __forceinline__ float fy_load24(const float* __restrict__ base3, int64_t entry) {
    const uint8_t* __restrict__ b = reinterpret_cast<const uint8_t*>(base3) + entry * 3;
    const uint32_t v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
    return __uint_as_float(v << FY_P24_SHIFT);
}
__global__ __launch_bounds__(256) void k_bound_repair(RepairArgs A) {
    // One WORKGROUP per user: its four waves take every fourth surviving block (a wave per user left a cluster of 800 users on 200
    // workgroups, each walking its user's blocks one after the other: 0.64 ms per cluster at 200 clusters), the new masks meet in
    // LDS and wave 0 compacts the user's list.
    __shared__ uint8_t sh_newmask[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double LN2 = 0.69314718055994530942;
    unsigned long long dropped = 0;
    for (int u = blockIdx.x; u < A.n_users; u += gridDim.x) {       // block-uniform
        const int ns = A.n_surv[u];
        const int slot = A.slot0 + u;
        const int r0 = A.rowptr[slot], r1 = A.rowptr[slot + 1];
        if (ns == 0 || r1 - r0 > REPAIR_MAX_LIST) continue;
        uint16_t* __restrict__ mine = A.surv + (int64_t)u * A.ldsurv;
        uint8_t* __restrict__ mask_of = A.surv_mask + (int64_t)u * A.ldsurv;
        const float t = A.tau[u];
        const double base = A.pvpi[slot - A.slot_lo];
        for (int k0 = 0; k0 < ns; k0 += 1024) {                     // (a list longer than the LDS array: in pieces)
            const int nk = min(1024, ns - k0);
            for (int k = k0 + wave; k < k0 + nk; k += 4) {          // wave-uniform
                const int blk = (int)mine[k];
                unsigned mask = mask_of[k];
                for (int qb = 0; qb < 4; qb++) {
                    if (!((mask >> qb) & 1u)) continue;
                    const int sb = 4 * blk + qb;
                    const int c0 = sb * 64;
                    // the user's items inside the sub-block: the largest x among them (0: none -- the bound stands)
                    float ymax = 0.f;
                    for (int f = r0 + lane; f < r1; f += 64) {
                        const unsigned d = (unsigned)(A.csr_idx[f] - c0);
                        if (d < 64u) ymax = fmaxf(ymax, A.csr_x[f]);
                    }
                    for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
                    if (ymax == 0.f) continue;
                    const float am = A.amax64[sb], bm = A.bmax64[sb];
                    double sum = 0.0;
                    for (int f = r0 + lane; f < r1; f += 64) {
                        const int j = A.csr_idx[f];
                        float g = fy_load24(A.Bmax64, (int64_t)j * A.ldb64 + sb);
                        if (g > 0.f) {
                            if (j >= A.p_eff && c0 >= A.p_eff) {
                                g = fmaxf(0.f, g - A.w2 * A.csr_x[f] * ymax * 0.999999f);
                            } else {
                                const uint32_t rep = A.Brep[(int64_t)j * A.ldb64 + sb];
                                const int col = c0 + (int)(rep & 63u);
                                int l = r0, h = r1;                     // did the user rate the column of the maximum?
                                while (l < h) { const int m = (l + h) >> 1; if (A.csr_idx[m] < col) l = m + 1; else h = m; }
                                if (l < r1 && A.csr_idx[l] == col) g = __uint_as_float((rep >> 8) << FY_P24_SHIFT);
                            }
                        }
                        sum += (double)fy_log2(fmaf(A.csr_q[f], bm, fmaf(am, A.csr_e[f], g)));
                    }
                    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
                    if (!fy_bound_keeps((float)(base + LN2 * sum), t, (float)base)) { mask &= ~(1u << qb); dropped++; }
                }
                if (lane == 0) sh_newmask[k - k0] = (uint8_t)mask;
            }
            __syncthreads();
            if (wave == 0) {                                            // compaction in place (kept <= position read)
                int kept = k0 == 0 ? 0 : A.n_surv[u];                   // (pieces behind the first append to what the earlier ones kept)
                for (int kk = 0; kk < nk; kk += 64) {
                    const int k = kk + lane;
                    const unsigned m = k < nk ? sh_newmask[k] : 0u;
                    const int blk = k < nk ? (int)mine[k0 + k] : 0;
                    const unsigned long long bal = __ballot(m != 0u);
                    if (m) {
                        const int at = kept + __popcll(bal & ((1ull << lane) - 1ull));
                        mine[at] = (uint16_t)blk;
                        mask_of[at] = (uint8_t)m;
                    }
                    kept += __popcll(bal);
                }
                if (lane == 0) A.n_surv[u] = kept;
            }
            __syncthreads();
        }
    }
    if (lane == 0 && dropped) atomicAdd(&A.counters[4], dropped);
}

// Exact scores of the surviving blocks that lie behind the column panel ("strays"), from the sparse data alone:
//     G[j][i] = (1-l)^2 sum_{v in raters(i)} x_vi x_vj     for the user's rated items j and the candidates i of the block.
// Who they are (measured, ML-25M shape in 50 clusters): blocks of users with 15-50 ratings around the user's OWN rated tail
// items -- G[j][i0] of a rated tail item i0 holds the user's own large x_ui0 x_uj for every j of the list, so the maxima of
// i0's 64-column sub-block bound every term high, while the 63 real candidates next to i0 score ~100 below the threshold.
// One workgroup per user with strays:
//   (1) the co-rater table T of the user: every (rater v, list position of j, x_vj) over the raters of the user's items j,
//       grouped by v with a counting sort in LDS (the CSC lists of the j are streamed once, coalesced);
//   (2) per stray block, per live candidate i (a wave each): for every rater v of i, T's entries of v are exactly the
//       co-ratings of v with the user's list: x_vi x_vj is added to the accumulator of j's list position (fixed point,
//       ds_add_u64: the sum does not depend on the order of the lanes), then the log terms are summed over the list.
// A candidate costs its co-ratings with the list, not the length of its raters' rows.  Lists longer than STRAY_JCAP (or with
// more than STRAY_TCAP co-raters) are walked in pieces -- the score is a sum over j -- and T is then rebuilt per block.
constexpr int STRAY_UCAP = 32768;    // users of the cluster (panel mode is for many small clusters: fy_rm2.hip, Plan::panel)
constexpr int STRAY_JCAP = 512;
constexpr int STRAY_TCAP = 65536;    // >= STRAY_UCAP: one item's raters always fit
struct StrayArgs {
    const int32_t* __restrict__ surv_prefix;
    const uint16_t* __restrict__ surv;
    const uint8_t* __restrict__ surv_mask;
    int64_t ldsurv;
    int32_t n_users, slot0, slot_lo, slot_base, Uc, panel_blocks, Ic, pbase;
    const int32_t* __restrict__ rowptr;
    const int32_t* __restrict__ csr_idx;
    const float* __restrict__ csr_e;
    const float* __restrict__ csr_q;
    const int32_t* __restrict__ rank_pair;
    const int32_t* __restrict__ pair_start;
    const int32_t* __restrict__ csc_slot;
    const float* __restrict__ csc_x;
    const float* __restrict__ a_rank;
    const float* __restrict__ b_rank;
    const double* __restrict__ pvpi;
    float w2;
    float* __restrict__ S;
    int2* __restrict__ T;                 // gridDim.x * STRAY_TCAP entries {list position, x_vj as bits}
    unsigned long long* __restrict__ counters;
    const int2* __restrict__ items;       // k_stray_items: {user of the batch, index of its first stray block of this item}
    const int32_t* __restrict__ n_items;
};
// Work items of k_score_stray: a user's stray blocks in groups of STRAY_GROUP (a user has 4 on average and up to ~20: one
// workgroup per USER left the launch waiting for the users with the most blocks; the co-rater table is rebuilt per group).
// One block per item since round 3: a cluster's launch has a few hundred items for 1 500 work-group slots and lasts as long as its
// slowest item -- groups of four: 200 clusters of ML-25M shape 345 ms per job, single blocks 285 ms.
constexpr int STRAY_GROUP = 1;
__global__ void k_stray_items(const int32_t* __restrict__ surv_prefix, const uint16_t* __restrict__ surv, int64_t ldsurv, int32_t n_users,
                              int32_t panel_blocks, int2* __restrict__ items, int32_t* __restrict__ n_items) {
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n_users; u += gridDim.x * blockDim.x) {
        const int ns = surv_prefix[u + 1] - surv_prefix[u];
        const uint16_t* __restrict__ mine = surv + (int64_t)u * ldsurv;
        int first = 0, hi = ns;                                   // the user's blocks are in ascending order: strays are a suffix
        while (first < hi) { const int mid = (first + hi) >> 1; if ((int)mine[mid] < panel_blocks) first = mid + 1; else hi = mid; }
        const int groups = (ns - first + STRAY_GROUP - 1) / STRAY_GROUP;
        if (groups > 0) {
            const int at = atomicAdd(n_items, groups);
            for (int g = 0; g < groups; g++) items[at + g] = make_int2(u, first + g * STRAY_GROUP);
        }
    }
}
extern __shared__ __attribute__((aligned(16))) int32_t fy_stray_off[];      // Uc + 1 rater offsets
__global__ __launch_bounds__(256) void k_score_stray(StrayArgs A) {
    int32_t* __restrict__ sh_off = fy_stray_off;
    __shared__ int32_t sh_col[STRAY_JCAP], sh_pre[STRAY_JCAP + 1], sh_q0[STRAY_JCAP];
    __shared__ float sh_e[STRAY_JCAP], sh_q[STRAY_JCAP];
    __shared__ unsigned long long sh_g[4][STRAY_JCAP];
    __shared__ double sh_t[PRUNE_BLOCK];       // log2 sums of the block's candidates; NaN = no candidate
    __shared__ int32_t sh_nk, sh_wsum[4];
    __shared__ int32_t sh_cq0[PRUNE_BLOCK], sh_cnr[PRUNE_BLOCK];      // the candidates' CSC ranges
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double LN2 = 0.69314718055994530942;
    // x_vi x_vj <= 1/4 (both ratings are part of s_v), so a sum over at most Uc = 2^15 raters stays below 2^13: 49 fraction bits
    const double fx = 562949953421312.0, fx_inv = 1.0 / 562949953421312.0;
    int2* __restrict__ T = A.T + (int64_t)blockIdx.x * STRAY_TCAP;
    unsigned long long my_terms = 0, my_blocks = 0;
    for (int t = threadIdx.x; t < 4 * STRAY_JCAP; t += blockDim.x) (&sh_g[0][0])[t] = 0ull;
    // entry e of the piece's concatenated rater lists -> (list position, CSC entry); four at a time, loads first
    auto locate = [&](int e, int nk, int& t, int& q) __attribute__((always_inline)) {
        int l = 0, h = nk;                                        // last t with sh_pre[t] <= e
        while (h - l > 1) { const int m = (l + h) >> 1; if (sh_pre[m] <= e) l = m; else h = m; }
        t = l;
        q = sh_q0[l] + (e - sh_pre[l]);
    };
    const int n_items = *A.n_items;
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {     // block-uniform
        const int u = A.items[it].x, first = A.items[it].y;
        const int wbase = A.surv_prefix[u], ns = min(A.surv_prefix[u + 1] - wbase, first + STRAY_GROUP);
        const uint16_t* __restrict__ mine = A.surv + (int64_t)u * A.ldsurv;
        const int slot = A.slot0 + u;
        const int r0 = A.rowptr[slot], r1 = A.rowptr[slot + 1];
        bool built = false, single = false;
        for (int ks = first; ks < ns; ks++) {
            const int w = wbase + ks;
            const int blk = (int)mine[ks];
            const unsigned sub_mask = A.surv_mask[(int64_t)u * A.ldsurv + ks];
            __syncthreads();                                        // the previous block's sh_t has been written out
            {   // candidates: dead (NaN) when out of range, in a sub-block whose bound stayed below the threshold, or rated by the user
                const int i = blk * PRUNE_BLOCK + threadIdx.x;
                bool dead = i >= A.Ic || !((sub_mask >> (threadIdx.x >> 6)) & 1u);
                if (!dead) {
                    int l = r0, h = r1;
                    while (l < h) { const int m = (l + h) >> 1; if (A.csr_idx[m] < i) l = m + 1; else h = m; }
                    dead = l < r1 && A.csr_idx[l] == i;
                }
                int cq0 = 0, cnr = 0;
                if (!dead) {
                    const int pr = A.rank_pair[A.pbase + i];
                    cq0 = A.pair_start[pr];
                    cnr = A.pair_start[pr + 1] - cq0;
                }
                sh_cq0[threadIdx.x] = cq0;
                sh_cnr[threadIdx.x] = cnr;                          // 0 raters: a dead candidate
                sh_t[threadIdx.x] = dead ? __builtin_nan("") : 0.0;
            }
            __syncthreads();                                        // a wave reads the flags of candidates other waves' threads set
            int k0 = r0;
            while (k0 < r1) {
                if (!(built && single)) {
                    // ---- (1) the piece of the list and its co-rater table
                    __syncthreads();                                // nobody reads the previous piece any more
                    const int ntry = min(STRAY_JCAP, r1 - k0);
                    for (int t = threadIdx.x; t < ntry; t += blockDim.x) {
                        const int j = A.csr_idx[k0 + t];
                        const int pr = A.rank_pair[A.pbase + j];
                        const int q0 = A.pair_start[pr];
                        sh_col[t] = j;
                        sh_e[t] = A.csr_e[k0 + t];
                        sh_q[t] = A.csr_q[k0 + t];
                        sh_q0[t] = q0;
                        sh_pre[t + 1] = A.pair_start[pr + 1] - q0;  // rater count; prefix below
                    }
                    for (int t = threadIdx.x; t <= A.Uc; t += blockDim.x) sh_off[t] = 0;
                    __syncthreads();
                    if (threadIdx.x == 0) {                         // as many items as fit the table (one always does)
                        int tot = 0, m = 0;
                        sh_pre[0] = 0;
                        while (m < ntry && tot + sh_pre[m + 1] <= STRAY_TCAP) { tot += sh_pre[m + 1]; sh_pre[m + 1] = tot; m++; }
                        sh_nk = m;
                    }
                    __syncthreads();
                    const int nk_ = sh_nk, tot = sh_pre[nk_];
                    for (int e0 = threadIdx.x; e0 < tot; e0 += 4 * 256) {          // count
                        int v[4];
#pragma unroll
                        for (int x = 0; x < 4; x++) {
                            const int e = e0 + x * 256;
                            v[x] = -1;
                            if (e < tot) { int t, q; locate(e, nk_, t, q); v[x] = A.csc_slot[q] - A.slot_base; }
                        }
#pragma unroll
                        for (int x = 0; x < 4; x++)
                            if (v[x] >= 0) atomicAdd(&sh_off[v[x] + 1], 1);
                    }
                    __syncthreads();
                    {   // inclusive scan of sh_off[1 .. Uc]: afterwards sh_off[v] = first entry of rater v, sh_off[Uc] = total
                        const int per = (A.Uc + (int)blockDim.x - 1) / (int)blockDim.x;
                        const int b0 = 1 + threadIdx.x * per, b1 = min(A.Uc + 1, b0 + per);
                        int sum = 0;
                        for (int t = b0; t < b1; t++) sum += sh_off[t];
                        int incl = sum;
                        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
                        if (lane == 63) sh_wsum[wave] = incl;
                        __syncthreads();
                        int before = incl - sum;
                        for (int x = 0; x < wave; x++) before += sh_wsum[x];
                        for (int t = b0; t < b1; t++) { before += sh_off[t]; sh_off[t] = before; }
                    }
                    __syncthreads();
                    // fill: the cursor of rater v is sh_off[v] (its start); afterwards sh_off[v] is its END and its start is
                    // sh_off[v - 1] (0 for the first)
                    for (int e0 = threadIdx.x; e0 < tot; e0 += 4 * 256) {
                        int v[4], tt[4];
                        float xv[4];
#pragma unroll
                        for (int x = 0; x < 4; x++) {
                            const int e = e0 + x * 256;
                            v[x] = -1;
                            tt[x] = 0;
                            xv[x] = 0.f;
                            if (e < tot) { int q; locate(e, nk_, tt[x], q); v[x] = A.csc_slot[q] - A.slot_base; xv[x] = A.csc_x[q]; }
                        }
#pragma unroll
                        for (int x = 0; x < 4; x++)
                            if (v[x] >= 0) T[atomicAdd(&sh_off[v[x]], 1)] = make_int2(tt[x], __float_as_int(xv[x]));
                    }
                    __threadfence_block();
                    __syncthreads();
                    built = true;
                    single = (k0 == r0 && nk_ == r1 - r0);
                }
                const int nk = sh_nk;
                // ---- (2) the candidates of this block against the piece
                unsigned long long* __restrict__ g = sh_g[wave];
                // the wave's live candidates, one after the other; the first 64 raters of the NEXT one are loaded while the
                // current one's table entries are fetched (a candidate is two dependent round trips otherwise)
                auto next_live = [&](int c) __attribute__((always_inline)) {
                    while (c < PRUNE_BLOCK && sh_cnr[c] == 0) c += 4;
                    return c;
                };
                int c = next_live(wave);
                int pv = 0;
                float px = 0.f;
                if (c < PRUNE_BLOCK && lane < sh_cnr[c]) { pv = A.csc_slot[sh_cq0[c] + lane] - A.slot_base; px = A.csc_x[sh_cq0[c] + lane]; }
                while (c < PRUNE_BLOCK) {                           // wave-uniform
                    const int i = blk * PRUNE_BLOCK + c;
                    const int q0 = sh_cq0[c], nr = sh_cnr[c];
                    const int cn = next_live(c + 4);
                    int v = pv;
                    float xf = px;
                    if (cn < PRUNE_BLOCK && lane < sh_cnr[cn]) { pv = A.csc_slot[sh_cq0[cn] + lane] - A.slot_base; px = A.csc_x[sh_cq0[cn] + lane]; }
                    for (int qb = 0; qb < nr; qb += 64) {           // a lane per rater of the candidate
                        if (qb > 0 && qb + lane < nr) { v = A.csc_slot[q0 + qb + lane] - A.slot_base; xf = A.csc_x[q0 + qb + lane]; }
                        if (qb + lane < nr) {
                            const double xi = (double)xf;
                            const int t0 = v ? sh_off[v - 1] : 0, t1 = sh_off[v];
                            for (int t = t0; t < t1; t += 4) {      // four table entries in flight
                                int2 ent[4];
#pragma unroll
                                for (int x = 0; x < 4; x++) ent[x] = T[min(t + x, t1 - 1)];
#pragma unroll
                                for (int x = 0; x < 4; x++)
                                    if (t + x < t1) atomicAdd(&g[ent[x].x], (unsigned long long)(xi * (double)__int_as_float(ent[x].y) * fx + 0.5));
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const float ai = A.a_rank[i], bi = A.b_rank[i];
                    double tt = 0.0;
                    for (int k = lane; k < nk; k += 64) {
                        const float gv = A.w2 * (float)((double)g[k] * fx_inv);
                        {
                            float lm = 1.f;
                            int le = 0;
                            fy_logprod_step(fmaf(sh_q[k], bi, fmaf(ai, sh_e[k], gv)), lm, le);
                            tt += fy_logprod_fold(lm, le);
                        }
                        g[k] = 0ull;
                    }
                    for (int o = 32; o > 0; o >>= 1) tt += __shfl_xor(tt, o, 64);
                    if (lane == 0) sh_t[c] += tt;                   // (candidate c belongs to this wave alone)
                    c = cn;
                }
                k0 += nk;
            }
            __syncthreads();
            const double base = A.pvpi[slot - A.slot_lo];
            const double tv = sh_t[threadIdx.x];
            A.S[(int64_t)w * PRUNE_BLOCK + threadIdx.x] = tv != tv ? __builtin_nanf("") : (float)(base + LN2 * tv);
            if (threadIdx.x == 0) {
                my_terms += (unsigned long long)(r1 - r0) * 64ull * (unsigned long long)__popc(sub_mask);
                my_blocks++;
            }
        }
    }
    if (threadIdx.x == 0 && my_blocks) {
        atomicAdd(&A.counters[1], my_terms);
        atomicAdd(&A.counters[3], my_blocks);
    }
}

// ================================================================ cooperative ranks (fy_collectives): kernels
// A cooperative rank owns the item rows  me, me + world, me + 2 world, ...  (rows are in popularity order, so every rank
// gets the same mix of heavy and light rows -- no work model needed).  Its view of the users is a compact CSR that keeps only
// the rated items falling into its rows, renumbered to the LOCAL row index (idx / world): one wave per user, ballot
// compaction, order preserved.
__global__ void k_my_csr_count(int32_t n_slots, int32_t slot_base, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr_idx,
                               int32_t world, int32_t me, int32_t* __restrict__ cnt) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int32_t v = blockIdx.x * wpb + (threadIdx.x >> 6); v <= n_slots; v += gridDim.x * wpb) {
        int c = 0;
        if (v < n_slots)
            for (int32_t f = rowptr[slot_base + v] + lane; f < rowptr[slot_base + v + 1]; f += 64) c += (csr_idx[f] % world) == me;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if (lane == 0) cnt[v] = c;     // cnt[n_slots] = 0: the exclusive scan's last element is the total
    }
}
__global__ void k_my_csr_fill(int32_t n_slots, int32_t slot_base, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr_idx,
                              const float* __restrict__ csr_e, const float* __restrict__ csr_q, int32_t world, int32_t me,
                              const int32_t* __restrict__ my_rowptr, int32_t* __restrict__ my_idx, float* __restrict__ my_e,
                              float* __restrict__ my_q) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int32_t v = blockIdx.x * wpb + (threadIdx.x >> 6); v < n_slots; v += gridDim.x * wpb) {
        int at = my_rowptr[v];
        const int32_t a = rowptr[slot_base + v], b = rowptr[slot_base + v + 1];
        for (int32_t f0 = a; f0 < b; f0 += 64) {
            const int32_t f = f0 + lane;
            const int32_t j = f < b ? csr_idx[f] : -1;
            const bool mine = j >= 0 && (j % world) == me;
            const unsigned long long bal = __ballot(mine);
            if (mine) {
                const int k = at + __popcll(bal & ((1ull << lane) - 1ull));
                my_idx[k] = j / world;
                my_e[k] = csr_e[f];
                my_q[k] = csr_q[f];
            }
            at += __popcll(bal);
        }
    }
}

// CSC entries (rater slot, weight) of item rows r0, r0 + stride, ... (nrows of them), row by row: the compact CSC a cooperative rank builds its
// segment table from
__global__ void k_gather_rows(int32_t r0, int32_t stride, int32_t nrows, const int32_t* __restrict__ rank_pair, const int32_t* __restrict__ pair_start,
                              const int32_t* __restrict__ local_start, const int32_t* __restrict__ csc_slot,
                              const float* __restrict__ csc_w, int32_t* __restrict__ my_slot, float* __restrict__ my_w) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int32_t i = blockIdx.x * wpb + (threadIdx.x >> 6); i < nrows; i += gridDim.x * wpb) {
        const int32_t pr = rank_pair[r0 + i * stride];
        const int32_t q0 = pair_start[pr], n = pair_start[pr + 1] - q0, l0 = local_start[i];
        for (int32_t k = lane; k < n; k += 64) { my_slot[l0 + k] = csc_slot[q0 + k]; my_w[l0 + k] = csc_w[q0 + k]; }
    }
}

// owner of the user: blocks behind the seed whose (summed) bound reaches tau_u, in ascending block order
// ---- super-block bounds (one-cluster pruned job).  Super-block s < n_sup covers the fine 256-column blocks [first[s], first[s + 1]):
// Bsup[j][s] = the largest entry of matrix row j in those blocks (= max of the Bmax entries: exact, the maximum of 24-bit values is
// one), asup / bsup_b likewise from the blocks' maxima of a and b.  A bound over a super-block is >= the bound of each of its blocks,
// so a super-block below tau_u excludes all of them; one that is kept hands ALL its fine blocks to the survivor pass.
// (Measured at ML-25M shape, N = 50: the 22 600 surviving (user, block) pairs of the headline job lie in blocks 1 .. 8 of 231.)
__global__ void k_build_bsup(int32_t Ic, int32_t n_sup, const int32_t* __restrict__ first, const float* __restrict__ Bmax, int64_t ldb,
                             float* __restrict__ Bsup) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int32_t j = blockIdx.x * wpb + (threadIdx.x >> 6); j < Ic; j += gridDim.x * wpb) {
        float m = 0.f;
        if (lane < n_sup)
            for (int b = first[lane]; b < first[lane + 1]; b++) m = fmaxf(m, fy_load24(Bmax, (int64_t)j * ldb + b));
        Bsup[(int64_t)j * 64 + lane] = m;
    }
}
__global__ void k_sup_amax(int32_t n_sup, const int32_t* __restrict__ first, const float* __restrict__ amax, const float* __restrict__ bmax,
                           float* __restrict__ asup, float* __restrict__ bsup_b) {
    const int s = threadIdx.x;
    if (s >= 64) return;
    float ma = 0.f, mb = 0.f;
    if (s < n_sup)
        for (int b = first[s]; b < first[s + 1]; b++) { ma = fmaxf(ma, amax[b]); mb = fmaxf(mb, bmax[b]); }
    asup[s] = ma;
    bsup_b[s] = mb;
}
// survivors from the super-block bounds: one wave per user, the fine blocks of every kept super-block in ascending order
__global__ __launch_bounds__(256) void k_bound_select_sup(const float* __restrict__ UBsup, int32_t n_sup, const int32_t* __restrict__ first,
                                                          const float* __restrict__ tau, const double* __restrict__ pvpi /* [u] */, int32_t n_users,
                                                          int64_t ldb, uint16_t* __restrict__ surv, int32_t* __restrict__ n_surv) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int u = blockIdx.x * wpb + (threadIdx.x >> 6); u < n_users; u += gridDim.x * wpb) {
        const float t = tau[u];
        const float pv = (float)pvpi[u];
        // (tau = +inf: the user emits nothing and its UBsup row was never written)
        const bool keep = lane < n_sup && t != INFINITY && fy_bound_keeps(UBsup[(int64_t)u * 64 + lane], t, pv);
        const int width = keep ? first[lane + 1] - first[lane] : 0;
        int incl = width;                                   // inclusive prefix of the widths over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int x = __shfl_up(incl, o, 64);
            if (lane >= o) incl += x;
        }
        const int at = incl - width;
        for (int k = 0; k < width; k++) surv[(int64_t)u * ldb + at + k] = (uint16_t)(first[lane] + k);
        const int total = __shfl(incl, 63, 64);
        if (lane == 0) n_surv[u] = total;
    }
}
__global__ __launch_bounds__(256) void k_bound_select(const float* __restrict__ UB, int64_t ldb, int32_t nblk, int32_t seed_blocks,
                                                      const float* __restrict__ tau, const double* __restrict__ pvpi /* [u] */,
                                                      int32_t n_users, uint16_t* __restrict__ surv, int32_t* __restrict__ n_surv) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int u = blockIdx.x * wpb + (threadIdx.x >> 6); u < n_users; u += gridDim.x * wpb) {
        const float t = tau[u];
        const float pv = (float)pvpi[u];
        int count = 0;
        for (int b0 = 0; b0 < nblk; b0 += 64) {
            const int b = b0 + lane;
            const bool keep = b < nblk && b >= seed_blocks && fy_bound_keeps(UB[(int64_t)u * ldb + b], t, pv);
            const unsigned long long bal = __ballot(keep);
            if (keep) surv[(int64_t)u * ldb + count + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)b;
            count += __popcll(bal);
        }
        if (lane == 0) n_surv[u] = count;
    }
}

// (slot, block) of every surviving block of this rank's users, packed for the all-gather
// ================================================================ refinement of ill-conditioned list rows (round 4)
// score(u, i) = pvpi + sum of ~n log terms of magnitude ~10 each; where the two cancel (|score| ~ 1: users with a dozen ratings, every
// multi-cluster job) the absolute error of the production arithmetic -- 17 mantissa bits per matrix element -- is a RELATIVE error of
// the same size: the one row that carried the all-rows maximum against the fp64 definition (6.3e-6 of north_star's 1e-5) was such a
// row.  After a cluster's lists stand, every list row with |score| < refine_c * sqrt(n) whose item lies in the first 256 columns (the
// popular candidates: where list items are) is scored AGAIN in fp64 from the UNROUNDED matrix values the row kernel's epilogue kept for
// exactly those columns (MEpilogue::head32 / tail32: fp64 images of the exact fixed-point sums) and the fp64 statistics (p, b, s_u):
//     G(j, c) = head32[min(j, c)][max(j, c)]        (row min(j, c) < 256 walked column max(j, c): G is symmetric)
//             = tail32[j - tail_from][c]            symmetric panel mode, j a tail row (its walk covered the head columns)
// and the user's list is put back in order (a re-scored row moves by ~1e-6 of its value).  n gathers and n fp64 logs per row.
__global__ void k_refine_colmap(int32_t n_cols, const int32_t* __restrict__ rank_item_raw /* offset by pbase */, int32_t* __restrict__ colmap) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_cols) colmap[rank_item_raw[c]] = c;
}
struct RefineArgs {
    int32_t slot0, n_users, slot_lo, slot_base;
    const int32_t* __restrict__ n_out;       // by slot - slot_lo
    const int32_t* __restrict__ out_off;
    const int32_t* __restrict__ out_item;    // raw ids
    float* __restrict__ out_score;
    int32_t* __restrict__ out_item_w;        // (the same array, written when a list is re-sorted)
    const int32_t* __restrict__ rowptr;      // global, by slot
    const int32_t* __restrict__ csr_idx;
    const float* __restrict__ csr_r;
    const double* __restrict__ usum_slot;
    const double* __restrict__ p_rank;       // offset by pbase
    const double* __restrict__ b_rank;
    const int32_t* __restrict__ colmap;      // raw item id -> column (< n_cols) or -1
    int32_t max_item;
    const double* __restrict__ head32;
    int64_t ld_head;
    const double* __restrict__ tail32;
    int32_t tail_from, head_rows;            // head_rows: rows of head32 = columns of tail32 = columns the pass covers
    double unscale;                          // 1 / (the 2^-c the stored matrix is scaled by)
    double lambda, ln_items, ln_users;       // ln(numberOfItems), ln(U_c)
    double users_minus_1;
    float refine_c;
    unsigned long long* __restrict__ n_refined;
};
__global__ __launch_bounds__(256) void k_refine_rows(RefineArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint64_t fy_refine_keys[];      // [4 waves][lp2]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int u = blockIdx.x * 4 + w;
    if (u >= A.n_users) return;                                   // wave-uniform; nothing below synchronises the workgroup
    const int slot = A.slot0 + u;
    const int K = A.n_out[slot - A.slot_lo];
    if (K == 0) return;
    const int beg = A.rowptr[slot], end = A.rowptr[slot + 1], n = end - beg;
    const int off = A.out_off[slot - A.slot_lo];
    const float thr = A.refine_c * sqrtf((float)n);
    const double su = A.usum_slot[slot];
    const double l = A.lambda;
    int changed = 0;
    for (int base = 0; base < K; base += 64) {                    // 64 list rows per trip: one load per lane, the rows to re-score by ballot
        const int rl = base + lane;
        int cl = -1;
        if (rl < K) {
            const float s = A.out_score[off + rl];
            if (fabsf(s) < thr) {                                 // (also false for -inf / NaN)
                const int raw = A.out_item[off + rl];
                if (raw >= 0 && raw <= A.max_item) cl = A.colmap[raw];
            }
        }
        unsigned long long todo = __ballot(cl >= 0);
        while (todo) {                                            // wave-uniform
            const int bsel = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int r = base + bsel;
            const int c = __shfl(cl, bsel, 64);
            const double pc = A.p_rank[c], bc = A.b_rank[c];
            double sum = 0.0;
            for (int k = lane; k < n; k += 64) {
                const int j = A.csr_idx[beg + k];
                const double x = (double)A.csr_r[beg + k] / su;
                double g32;
                if (A.tail32 && j >= A.tail_from) g32 = A.tail32[(int64_t)(j - A.tail_from) * A.head_rows + c];
                else g32 = A.head32[(int64_t)min(j, c) * A.ld_head + max(j, c)];
                const double pj = A.p_rank[j];
                const double e = (1.0 - l) * (A.b_rank[j] - x) + l * A.users_minus_1 * pj;
                const double term = g32 * A.unscale + l * (1.0 - l) * pj * bc + l * pc * e;
                sum += log(term);
            }
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            const double score = (double)(n - 1) * A.ln_items - (double)n * A.ln_users + sum;
            if (lane == 0) A.out_score[off + r] = (float)score;
            changed++;
        }
    }
    if (!changed) return;
    if (lane == 0 && A.n_refined) atomicAdd(A.n_refined, (unsigned long long)changed);
    // back in order: descending score, ties by ascending raw item id (the top-N kernels' key)
    int lp2 = 64;
    while (lp2 < K) lp2 <<= 1;
    uint64_t* __restrict__ keys = fy_refine_keys + (size_t)w * lp2;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");        // lane 0's stores above are visible to the wave's loads
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < lp2; i += 64) {
        uint64_t kk = 0ull;
        if (i < K) {
            // (agent-scope load: lane 0's re-scored values must not come from a stale L1 line)
            const uint32_t sb = __hip_atomic_load(reinterpret_cast<const uint32_t*>(A.out_score) + off + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            kk = ((uint64_t)fy_order_key(__uint_as_float(sb)) << 32) | (uint32_t)(0x7FFFFFFF - A.out_item[off + i]);
        }
        keys[i] = kk;
    }
    fy_wave_fence();
    for (int k = 2; k <= lp2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < lp2; i += 64) {
                const int x = i ^ j;
                if (x > i) {
                    const uint64_t a = keys[i], b = keys[x];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[x] = a; }
                }
            }
            fy_wave_fence();
        }
    for (int i = lane; i < K; i += 64) {
        const uint64_t c = keys[i];
        A.out_item_w[off + i] = 0x7FFFFFFF - (int32_t)(uint32_t)c;
        A.out_score[off + i] = fy_order_unkey((uint32_t)(c >> 32));
    }
}

// how many users keep block b (debug: FY_DEBUG_SYNC=3 prints the distribution; round 4 measured it to decide on a lazy mirror pass)
__global__ void k_surv_block_counts(int32_t n_users, const int32_t* __restrict__ n_quads, const uint16_t* __restrict__ surv, int64_t ldb, int32_t* __restrict__ counts) {
    for (int32_t u = blockIdx.x; u < n_users; u += gridDim.x)
        for (int k = threadIdx.x; k < n_quads[u]; k += blockDim.x) atomicAdd(&counts[surv[(int64_t)u * ldb + k]], 1);
}
__global__ void k_surv_entries(int32_t n_users, int32_t slot0, const int32_t* __restrict__ prefix, const uint16_t* __restrict__ surv,
                               int64_t ldb, long long* __restrict__ entries) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int u = blockIdx.x * wpb + (threadIdx.x >> 6); u < n_users; u += gridDim.x * wpb) {
        const int a = prefix[u], n = prefix[u + 1] - a;
        for (int k = lane; k < n; k += 64)
            entries[a + k] = ((long long)(slot0 + u) << 16) | (long long)surv[(int64_t)u * ldb + k];
    }
}

// partial exact scores of the surviving blocks of ALL ranks over this rank's item rows: one wave = one entry,
// 256 floats at Spart[w * 256]; entry w = k * t_max + i belongs to rank k and exists when i < counts[k]
template <int SB>
__global__ __launch_bounds__(256) void k_score_entries(const float* __restrict__ M_, const float* __restrict__ a_rank_,
                                                       const float* __restrict__ b_rank_,
                                                       const int32_t* __restrict__ range_off_, const int32_t* __restrict__ csr_idx_,
                                                       const float* __restrict__ csr_e_, const float* __restrict__ csr_q_,
                                                       const double* __restrict__ pvpi_,
                                                       const long long* __restrict__ entries_, const int32_t* __restrict__ counts_,
                                                       int32_t world, int32_t t_max, int32_t slot_base, int32_t Ic, int64_t ldm,
                                                       int32_t row_mul, int32_t row_add, float* __restrict__ Spart_,
                                                       unsigned long long* __restrict__ counters) {
    const int lane = threadIdx.x & 63;
    const int wave_in_grid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    const int total = world * t_max;
    const int64_t pitch = ldm * 3;
    const double LN2 = 0.69314718055994530942;
    const float qnan = __builtin_nanf("");
    unsigned long long my_terms = 0;
    for (int w = wave_in_grid; w < total; w += n_waves) {
        const int k = w / t_max;
        if (w - k * t_max >= counts_[k]) continue;
        const long long en = entries_[w];
        const int slot = (int)(en >> 16);
        const int col0 = (int)(en & 0xFFFF) * PRUNE_BLOCK;
        const int col = col0 + lane * 4;
        float a[4], bb[4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            a[v] = col + v < Ic ? a_rank_[col + v] : 0.0f;
            bb[v] = col + v < Ic ? b_rank_[col + v] : 0.0f;
        }
        const char* __restrict__ Mcol = reinterpret_cast<const char*>(M_) + (int64_t)col * 3;
        const int beg = range_off_[slot - slot_base], end = range_off_[slot - slot_base + 1];
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        unsigned mask = 0;
        // (idx, e, q) of a batch by three vector loads, one batch ahead, handed out with v_readlane: see k_score's walk
        const int sub = lane & (SB - 1);
        int vi = 0, vi_n;
        float ve = 0.f, vq = 0.f, ve_n, vq_n;
        if (beg < end) {
            const int kk = min(beg + sub, end - 1);
            vi = csr_idx_[kk]; ve = csr_e_[kk]; vq = csr_q_[kk];
        }
        for (int kk0 = beg; kk0 < end; kk0 += SB) {
            U3 g[SB];
            float e[SB], qq[SB];
            int jj[SB];
#pragma unroll
            for (int x = 0; x < SB; x++) {
                jj[x] = __builtin_amdgcn_readlane(vi, x);
                e[x] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ve), x));
                qq[x] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vq), x));
                g[x] = *reinterpret_cast<const U3*>(Mcol + (int64_t)jj[x] * pitch);
            }
            {
                const int kk = min(kk0 + SB + sub, end - 1);
                vi_n = csr_idx_[kk]; ve_n = csr_e_[kk]; vq_n = csr_q_[kk];
            }
            float p[4] = {1.f, 1.f, 1.f, 1.f};     // (product of the mantissas in fp32, exponents exact: see k_score's walk)
            int pe[4] = {0, 0, 0, 0};
#pragma unroll
            for (int x = 0; x < SB; x++) {
                if (kk0 + x < end) {
                    float gv[4];
                    fy_unpack24(g[x], gv);
#pragma unroll
                    for (int v = 0; v < 4; v++) fy_logprod_step(fmaf(qq[x], bb[v], fmaf(a[v], e[x], gv[v])), p[v], pe[v]);
                    const unsigned d = (unsigned)(jj[x] * row_mul + row_add - col);
                    if (d < 4u) mask |= 1u << d;
                }
            }
#pragma unroll
            for (int v = 0; v < 4; v++) t[v] += fy_logprod_fold(p[v], pe[v]);
            vi = vi_n; ve = ve_n; vq = vq_n;
        }
        const double base = pvpi_[slot - slot_base];
        float4 o;
        o.x = ((mask & 1u) || col + 0 >= Ic) ? qnan : (float)(base + LN2 * t[0]);
        o.y = ((mask & 2u) || col + 1 >= Ic) ? qnan : (float)(base + LN2 * t[1]);
        o.z = ((mask & 4u) || col + 2 >= Ic) ? qnan : (float)(base + LN2 * t[2]);
        o.w = ((mask & 8u) || col + 3 >= Ic) ? qnan : (float)(base + LN2 * t[3]);
        *reinterpret_cast<float4*>(Spart_ + (int64_t)w * PRUNE_BLOCK + lane * 4) = o;
        my_terms += (unsigned long long)(end - beg) * 256ull;
    }
    if (lane == 0 && my_terms) atomicAdd(&counters[1], my_terms);
}

