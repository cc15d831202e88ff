// fy_cooc.hpp -- the co-rating row kernel shared by both jobs.
//
// One workgroup owns one row `i` of the item x item co-rating matrix (restricted to a column chunk that fits LDS):
//     acc[j] = sum over users v who rated i of  w_vi * w_vj          (j in the chunk)
// walking the CSC column of i (who rated it) and, for every such user, the slice of the user's CSR row that falls
// into the chunk.  Accumulators are fp64 in LDS (ds_add_f64), so the summation order between waves changes the
// result only below 1e-15 relative.  The epilogue turns the finished row into either a dense row of RM2's M
// matrix (fy_rm2.hip) or the top-K similar items of item i (fy_itemsim.hip) -- nothing but the epilogue differs,
// which is the "same sparse co-rating Gram" observation of SURVEY.md section 8a.
//
// Replaces: the inner product  sum_v cache[v][i] * cache[v][j]  of M/rm/AbstractRM2Reducer.java:343-349 (hoisted out
// of the per-user loop) and Mahout RowSimilarityJob's CooccurrencesMapper / SimilarityReducer pair aggregation.
#pragma once
#include "fy_common.hpp"

namespace fy {

struct CoocArgs {
    // CSC of the cluster: pair_start indexed by pair id; rank_pair maps (pbase + row) -> pair id
    const int32_t* __restrict__ rank_pair;
    const int32_t* __restrict__ pair_start;
    const int32_t* __restrict__ csc_slot;
    const float* __restrict__ csc_w;
    // csc_slice[ch * nq + (e - q0)] = {first CSR entry, length} of the slice of rater e's CSR row that falls into column
    // chunk ch: precomputed per CSC entry so the row kernel reads it coalesced instead of gathering per-user offsets
    const int2* __restrict__ csc_slice;
    const int32_t* __restrict__ csr_idx;
    const float* __restrict__ csr_w;
    int32_t pbase;      // pcstart[c]
    int32_t slot_base;  // ucstart[c]
    int32_t Ic;         // items of the cluster
    int32_t CH;         // columns per chunk (LDS accumulators)
    int32_t nch;        // chunks per row
    int32_t row0;       // first row of this launch
    int32_t nrows;      // rows in this launch
    int32_t q0;         // first CSC entry of the cluster
    int32_t nq;         // CSC entries of the cluster
};

// Accumulates chunk `ch` of row `row` into the workgroup's dynamic LDS (fy_cooc_acc[0..CH), zeroed by the caller; all
// threads call this).
//
// A wave fetches the metadata of 64 of its raters at a time: lane l loads one rater's slot, weight and the [f0, f0+len)
// slice of that user's CSR row inside the chunk with two vector loads -- one dependent chain per 64 raters instead of
// one scalar chain per rater (the first version spent 88 % of its wave cycles waiting, rocprof r1a).  The wave then walks
// its raters four at a time (v_readlane broadcasts), issuing the four coalesced slice loads before the first LDS atomic
// so four memory round trips overlap.  The accumulators are addressed through the LDS symbol itself, so the
// compiler emits ds_add_f64 and knows they cannot alias the global arrays.
// (A fully load-balanced expansion with a per-lane binary search over the prefix sums was tried and ran 3x slower:
// six dependent ds_bpermute per step on a 16-wave workgroup.)
extern __shared__ double fy_cooc_acc[];

__device__ __forceinline__ void cooc_accumulate_row(const CoocArgs& A, int row, int ch) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int pair = A.rank_pair[A.pbase + row];
    const int e0 = A.pair_start[pair], e1 = A.pair_start[pair + 1];
    const int c0 = ch * A.CH;
    const int32_t* __restrict__ csr_idx = A.csr_idx;
    const float* __restrict__ csr_w = A.csr_w;
    // Rater (step s, slot q) of this wave = e0 + ((s * nwaves + wave) * RS + q): consecutive groups of RS raters go to
    // consecutive waves, so a row with few raters still spreads over the whole workgroup (one step per wave), while
    // lane l = RS s + q of the wave prefetches the metadata of 64 / RS steps at once.  (RS = 16 was slower: fewer
    // waves busy on short rows.)
    constexpr int RS = 4, STEPS = 64 / RS;
    for (int round = 0; e0 + round * nwaves * 64 < e1; round++) {
        const int s_l = lane / RS, q_l = lane % RS;
        const int e = e0 + ((round * STEPS + s_l) * nwaves + wave) * RS + q_l;
        int f0 = 0, len = 0;
        float w = 0.0f;
        if (e < e1) {
            w = A.csc_w[e];
            const int2 sl = A.csc_slice[(int64_t)ch * A.nq + (e - A.q0)];
            f0 = sl.x;
            len = sl.y;
        }
        const unsigned long long nonempty = __ballot(len > 0);
        for (int s = 0; s < STEPS; s++) {   // wave-uniform
            const unsigned long long rest = nonempty >> (RS * s);
            if (rest == 0) break;
            if ((rest & ((1ull << RS) - 1ull)) == 0) continue;
            int F[RS], L[RS];
            float W[RS];
#pragma unroll
            for (int q = 0; q < RS; q++) {
                F[q] = __builtin_amdgcn_readlane(f0, RS * s + q);
                L[q] = __builtin_amdgcn_readlane(len, RS * s + q);
                W[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w), RS * s + q));
            }
            int idx[RS];
            float x[RS];
#pragma unroll
            for (int q = 0; q < RS; q++) {
                idx[q] = 0; x[q] = 0.0f;
                if (lane < L[q]) { idx[q] = csr_idx[F[q] + lane]; x[q] = csr_w[F[q] + lane]; }
            }
#pragma unroll
            for (int q = 0; q < RS; q++)
                if (lane < L[q]) atomicAdd(&fy_cooc_acc[idx[q] - c0], (double)W[q] * (double)x[q]);
            // slices longer than one wave: heavy users, who are raters of very many rows -- half of all rater visits at
            // ML-25M shape.  Eight 64-entry segments are loaded before the first atomic so eight round trips overlap
            // (a plain one-segment loop here cost a full memory latency per 64 entries and dominated the kernel: 84 ->
            // 48 ms).  Two re-balancing schemes were tried and were slower: queueing the remainders in LDS and walking them
            // with the whole workgroup (104 ms: three barriers per 1024 raters), and giving segment s of rater r to wave
            // (r + s) mod nwaves with every wave walking all raters' metadata (90 ms: 16x the uncoalesced chunk_off
            // gathers).
#pragma unroll
            for (int q = 0; q < RS; q++) {
                for (int fb = 64; fb < L[q]; fb += 512) {   // wave-uniform
                    int ix[8];
                    float xx[8];
#pragma unroll
                    for (int z = 0; z < 8; z++) {
                        const int f = fb + 64 * z + lane;
                        ix[z] = 0; xx[z] = 0.0f;
                        if (f < L[q]) { ix[z] = csr_idx[F[q] + f]; xx[z] = csr_w[F[q] + f]; }
                    }
#pragma unroll
                    for (int z = 0; z < 8; z++)
                        if (fb + 64 * z + lane < L[q]) atomicAdd(&fy_cooc_acc[ix[z] - c0], (double)W[q] * (double)xx[z]);
                }
            }
        }
    }
}

// csc_slice table of one cluster from its chunk_off table (one thread per (chunk, CSC entry))
void build_csc_slices(Context* ctx, const int32_t* csc_slot, const int32_t* chunk_off, int32_t slot_base, int32_t q0, int32_t nq,
                      int32_t nch, int2* csc_slice, hipStream_t st = nullptr);

// chunk_off table for one cluster: one thread per (slot, boundary)
void build_chunk_offsets(Context* ctx, const int32_t* rowptr, const int32_t* csr_idx, int32_t slot_base, int32_t n_slots,
                         int32_t CH, int32_t nch, int32_t* chunk_off, hipStream_t st = nullptr);

// picks the chunk width for a cluster with Ic items: whole row when it fits the LDS budget
inline void pick_chunks(int32_t Ic, int32_t max_ch, int32_t& CH, int32_t& nch) {
    if (Ic <= max_ch) {
        CH = (int32_t)round_up(Ic > 0 ? Ic : 1, 64);
        nch = 1;
    } else {
        nch = (int32_t)ceil_div(Ic, max_ch);
        CH = (int32_t)round_up(ceil_div(Ic, nch), 64);
        nch = (int32_t)ceil_div(Ic, CH);
    }
}

}  // namespace fy
