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
    // CSR: chunk_off[(slot - slot_base) * (nch + 1) + ch] = first CSR entry of the slot's row with idx >= ch * CH
    const int32_t* __restrict__ chunk_off;
    const int32_t* __restrict__ csr_idx;
    const float* __restrict__ csr_w;
    int32_t pbase;      // pcstart[c]
    int32_t slot_base;  // ucstart[c]
    int32_t Ic;         // items of the cluster
    int32_t CH;         // columns per chunk (LDS accumulators)
    int32_t nch;        // chunks per row
    int32_t row0;       // first row of this launch
    int32_t nrows;      // rows in this launch
};

// Accumulates chunk `ch` of row `row` into acc[0..CH) (must be zeroed by the caller; all threads call this).
__device__ __forceinline__ void cooc_accumulate_row(const CoocArgs& A, int row, int ch, double* acc) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int pair = A.rank_pair[A.pbase + row];
    const int e0 = A.pair_start[pair], e1 = A.pair_start[pair + 1];
    const int c0 = ch * A.CH;
    const int stride = A.nch + 1;
    for (int e = e0 + wave; e < e1; e += nwaves) {
        const int v = A.csc_slot[e] - A.slot_base;
        const double wi = (double)A.csc_w[e];
        const int f0 = A.chunk_off[(int64_t)v * stride + ch];
        const int f1 = A.chunk_off[(int64_t)v * stride + ch + 1];
        for (int f = f0 + lane; f < f1; f += 64) {
            const int j = A.csr_idx[f] - c0;
            atomicAdd(&acc[j], wi * (double)A.csr_w[f]);   // ds_add_f64 (-munsafe-fp-atomics)
        }
    }
}

// chunk_off table for one cluster: one thread per (slot, boundary)
void build_chunk_offsets(Context* ctx, const int32_t* rowptr, const int32_t* csr_idx, int32_t slot_base, int32_t n_slots,
                         int32_t CH, int32_t nch, int32_t* chunk_off);

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
