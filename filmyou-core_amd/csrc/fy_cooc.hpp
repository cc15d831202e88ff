// fy_cooc.hpp -- the co-rating row kernel shared by both jobs.
//
// One workgroup owns one row `i` of the item x item co-rating matrix (restricted to a column chunk that fits LDS):
//     acc[j] = sum over users v who rated i of  w_vi * w_vj          (j in the chunk)
// walking the CSC column of i (who rated it) and, for every such user, the slice of the user's CSR row that falls
// into the chunk.  Accumulators live in LDS: 64-bit FIXED POINT (ds_add_u64) in the packed walk -- the product path of both
// jobs at the benchmark sizes: integer sums do not depend on the order of the atomics, the matrix is bit-reproducible -- and
// fp64 (ds_add_f64) where the ratings are not fp16-exact.  The epilogue turns the finished row into either a dense row of RM2's M
// matrix (fy_rm2.hip) or the top-K similar items of item i (fy_itemsim.hip) -- nothing but the epilogue differs,
// which is the "same sparse co-rating Gram" observation of SURVEY.md section 8a.
//
// Replaces: the inner product  sum_v cache[v][i] * cache[v][j]  of M/rm/AbstractRM2Reducer.java:343-349 (hoisted out
// of the per-user loop) and Mahout RowSimilarityJob's CooccurrencesMapper / SimilarityReducer pair aggregation.
#pragma once
#include <hip/hip_fp16.h>

#include <type_traits>

#include "fy_common.hpp"

namespace fy {

struct CoocArgs {
    // CSC of the cluster: pair_start indexed by pair id; rank_pair maps (pbase + row) -> pair id
    const int32_t* __restrict__ rank_pair;
    const int32_t* __restrict__ pair_start;
    // Segment table: the slice of every rater's CSR row that falls into column chunk ch is cut into segments of at most 64
    // entries.  seg_ptr[ch * (nq + 1) + (e - q0)] = first segment of CSC entry e in chunk ch (exclusive prefix, the
    // segments of one item row are contiguous); seg[k] = {first CSR entry, length}, seg_w[k] = the rater's weight.
    const int32_t* __restrict__ seg_ptr;
    const int2* __restrict__ seg;
    const float* __restrict__ seg_w;
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
    // optional (cooperative ranks): the segment table covers only the CSC entries of rows [row0, row0 + nrows), renumbered
    // row by row; local_start[k] = first local entry of the launch's k-th row (then q0 = 0, nq = local entries)
    const int32_t* __restrict__ local_start;
    int32_t row_stride;  // the launch's k-th row is row0 + k * row_stride (0 = 1): a cooperative rank owns rows r, r + world, ...
    // optional packed CSR (RM2 when every rating is exactly representable in fp16, e.g. half-star scales): 4 bytes per
    // entry = column index relative to its chunk (16 bits) | the raw rating as fp16; the rater's 1 / s_v is folded into
    // seg_w.  Halves the bytes of the stream that bounds the row kernel.
    const uint32_t* __restrict__ csr_pk;
    // optional: [k * nch + ch] = {first, end} segment of chunk ch of the launch's k-th row, precomputed (k_item_segments):
    // one load instead of the rank_pair -> pair_start -> seg_ptr chain in front of every (row, chunk) item
    const int2* __restrict__ item_seg;
    uint32_t pk_bytes;   // size of csr_pk in bytes (< 4 GiB): record count of the buffer descriptor the packed walk loads through
    // RM2 row kernel: explicit item list, item t = (row of the launch << 8) | chunk, with item_seg[t] its segment range.  In
    // the symmetric (half) walk a row has items only for its own chunk and the chunks behind it.
    const int32_t* __restrict__ item_id;
    int32_t half;        // 1: symmetric walk -- row i holds only the columns j > i, the mirror pass fills the rest
    // fixed-point accumulation (ACC = unsigned long long, packed walk only): a contribution is added as the integer
    // round(weight * rating * fx_scale), fx_scale = 2^k chosen per cluster so that no single contribution reaches 2^52 and no sum
    // 2^63 (fy_rm2.hip: fx_exponent).  ds_add_u64 is twice as fast as ds_add_f64 on gfx950 (10.8 against 20.9 cycles per wave
    // instruction at random addresses) and integer sums do not depend on the order of the atomics: the matrix is
    // bit-reproducible from run to run and from rank to rank.
    double fx_scale;
    // column-panel mode (fy_rm2.hip): the rows from tail_row0 on have items only for their first tail_chunks chunks
    // (tail_chunks = 0: every row has all its chunks)
    int32_t tail_row0, tail_chunks;
    // RM2 row kernel: consecutive items a workgroup takes from the item counter per atomic (0 = 1; fy_rm2.hip: cooc_item_grab)
    int32_t item_grab;
    // RM2 row kernel: the accumulators of a chunk are stored in four PLANES, column c at (c & 3) * acc_quarter + (c >> 2), so that
    // the epilogue -- every lane packs four consecutive columns -- reads consecutive 8-byte words from consecutive lanes (with
    // the columns stored linearly its four 32-byte-strided ds_read_b64 ran four ways bank-conflicted: SQ_LDS_BANK_CONFLICT was
    // 57 % of the LDS cycles of a 50-cluster job).  0 = linear (item-item similarity: its top-K epilogue walks single columns).
    int32_t acc_quarter;
    int32_t acc32;       // 1: 32-bit fixed-point accumulators (the launch picks the instantiation)
    // symmetric column-panel mode (fy_rm2.hip, Plan::psym): with `half`, only the rows in front of half_rows (the head rows = the
    // panel's columns) are cut symmetrically, and they have items only for their first head_chunks chunks -- the head rows' co-ratings
    // with the tail columns are the tail rows' with the head columns, which the tail rows store.  0 / 0 = plain half walk.
    int32_t half_rows, head_chunks;
};
// is row `row` of the launch walked symmetrically (only the columns behind it)?
__device__ __forceinline__ bool cooc_row_is_cut(const CoocArgs& A, int row) { return A.half && (A.half_rows == 0 || row < A.half_rows); }
__device__ __forceinline__ int cooc_acc_index(int c, int quarter) { return quarter ? (c & 3) * quarter + (c >> 2) : c; }

#ifndef FY_COOC_NB
#define FY_COOC_NB 8    // segment loads per group; two groups are in flight per wave (see cooc_accumulate_segments)
#endif
extern __shared__ __attribute__((aligned(16))) double fy_cooc_acc[];

// first batch of segment descriptors of this wave for the segment range [s_begin, s_end): issued early by the caller (the
// RM2 row kernel loads them for the NEXT item before the epilogue of the current one)
struct SegBatch {
    int2 d;
    float w;
};
// Which segment of an item's range a lane's descriptor slot holds.  The range is dealt out in GROUPS of 8 consecutive segments
// (the unit of the walk: 8 slice loads, then 8 LDS atomics), group G to wave G mod nwaves; a wave's `batch`-th vector load
// fetches the descriptors of 8 of its groups (lane l: group 8 batch + l / 8, segment l mod 8 of it).  Round 2 handed a wave 64
// CONSECUTIVE segments: fine for rows with thousands of segments, but in a cluster of 3 000 users (50 clusters at ML-25M shape)
// 83 % of the (row, chunk) items have at most 64 -- ONE wave of the workgroup walked them, eight groups one after the other,
// while the others waited at the barrier (13 us per item).  The valid slots of a batch are still a prefix of the lanes.
static_assert(FY_COOC_NB == 8, "the segment groups of cooc_seg_slot are 8 segments long");
__device__ __forceinline__ int cooc_seg_slot(int s_begin, int batch, int wave, int nwaves, int lane) {
    return s_begin + ((((batch << 3) + (lane >> 3)) * nwaves + wave) << 3) + (lane & 7);
}
__device__ __forceinline__ SegBatch cooc_first_batch(const CoocArgs& A, int s_begin, int s_end) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SegBatch B{make_int2(0, 0), 0.0f};
    const int s = cooc_seg_slot(s_begin, 0, wave, blockDim.x >> 6, lane);
    if (s < s_end) { B.d = A.seg[s]; B.w = A.seg_w[s]; }
    return B;
}

// The unpacked walk (8-byte CSR entries: ratings that are not fp16-exact).  ACC = double: ds_add_f64.  (ACC = float, ds_add_f32,
// is a MEASUREMENT build only, FY_COOC_F32: 193 cycles per wave instruction on gfx950 against 21 for ds_add_f64 -- 4x slower.)
template <bool PK, class ACC>
__device__ __forceinline__ void cooc_accumulate_segments(const CoocArgs& A, ACC* __restrict__ acc, int s_begin, int s_end, int c0, SegBatch first,
                                                         int rmask = -1 /* symmetric walk, own chunk: only columns > rmask (chunk-relative) count */) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int32_t* __restrict__ csr_idx = A.csr_idx;
    const float* __restrict__ csr_w = A.csr_w;
    const uint32_t* __restrict__ csr_pk = A.csr_pk;
    constexpr int NB = FY_COOC_NB;
    // A wave fetches 64 segment descriptors with one vector load and works through them in groups of NB: NB coalesced
    // slice loads, then NB LDS atomics.  Software-pipelined by hand (the compiler keeps the order it is given): the loads
    // of group g + 1 are issued BEFORE the atomics of group g, and the descriptors of the wave's next batch before the
    // first group of this one -- the kernel is bound by round trips to L2 / Infinity Cache per wave, not by bytes or by the
    // LDS atomics (a round-1 timing build without the atomics: 19.6 of 20.9 ms).
    struct Group {
        int L[NB];
        float W[NB];
        int idx[NB];
        float x[NB];
    };
    auto issue = [&](Group& G, const int2& d, float w, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const int F = __builtin_amdgcn_readlane(d.x, g + q);
            G.L[q] = __builtin_amdgcn_readlane(d.y, g + q);
            G.W[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w), g + q));
            // Unconditional loads, straight-line code: lanes beyond the segment re-read its first entry (same cache line,
            // no traffic; an empty descriptor reads entry 0).  With the loads inside `if (lane < L)` blocks the compiler's
            // wait-count pass gave up at the branches and drained ALL outstanding loads (s_waitcnt vmcnt(0)) before the
            // first use -- measured with a use inside the block: 42 ms instead of 24.
            const int at = F + (lane < G.L[q] ? lane : 0);
            G.x[q] = 0.0f;
            if constexpr (PK) G.idx[q] = (int)csr_pk[at];
            else {
                G.idx[q] = csr_idx[at] - c0;
                G.x[q] = csr_w[at];
            }
        }
    };
    auto commit = [&](Group& G) __attribute__((always_inline)) {
        if constexpr (PK) {
#pragma unroll
            for (int q = 0; q < NB; q++) {
                const uint32_t pk = (uint32_t)G.idx[q];
                G.x[q] = __half2float(__ushort_as_half((unsigned short)(pk >> 16)));
                G.idx[q] = (int)(pk & 0xFFFFu);
            }
        }
#pragma unroll
        for (int q = 0; q < NB; q++)
            if (lane < G.L[q] && G.idx[q] > rmask) atomicAdd(&acc[cooc_acc_index(G.idx[q], A.acc_quarter)], (ACC)G.W[q] * (ACC)G.x[q]);   // ds_add_f64 / ds_add_f32
    };
    int batch = 0;
    if (cooc_seg_slot(s_begin, 0, wave, nwaves, 0) >= s_end) return;
    int2 d = first.d;
    float w = first.w;
    while (true) {   // wave-uniform
        const int nloc = __popcll(__ballot(cooc_seg_slot(s_begin, batch, wave, nwaves, lane) < s_end));
        const int s_next = cooc_seg_slot(s_begin, batch + 1, wave, nwaves, lane);
        int2 dn = make_int2(0, 0);
        float wn = 0.0f;
        if (s_next < s_end) { dn = A.seg[s_next]; wn = A.seg_w[s_next]; }
        // Two groups of NB loads are in flight at any time.  The body is STRAIGHT-LINE code (one wave-uniform exit test per
        // pair of groups, no conditional issue): with `if (has_b) issue(...)` in the loop the compiler's wait-count pass merged
        // the two paths and waited for ALL outstanding loads (s_waitcnt vmcnt(0)) in front of every group of atomics, i.e. only
        // NB loads per wave were really in flight (ISA dump, round 2).  Descriptor slots behind `nloc` are empty (length 0,
        // first entry 0): their loads hit one cached line and their atomics are predicated off.
        Group GA, GB;
        issue(GA, d, w, 0);
#pragma unroll 1
        for (int g = 0; g < 64; g += 2 * NB) {
            issue(GB, d, w, g + NB);
            commit(GA);
            if (g + 2 * NB < 64) issue(GA, d, w, g + 2 * NB);      // (compile-time for the last trip of a 64-slot batch only when unrolled; cheap test otherwise)
            commit(GB);
            if (g + 2 * NB >= nloc) break;
        }
        d = dn;
        w = wn;
        batch++;
        if (cooc_seg_slot(s_begin, batch, wave, nwaves, 0) >= s_end) break;
    }
}

// The packed walk (PK), written for the instruction issue: round 2's counters showed the SIMDs issuing for 16.4 of the
// kernel's 17.8 ms (SQ_ACTIVE_INST_ANY; 21.8 VALU instructions per 64-entry segment), with the LDS at ~40 % and the memory
// system at ~60 % of what they sustain -- the walk is bound by instructions per segment, not by bytes, latency or atomics.
// Per segment now: three v_readlane (first entry, length, weight), ONE buffer_load_dword whose address needs no VALU at all
// (descriptor in SGPRs, the segment's first entry as scalar offset, lane * 4 as the constant vector offset; lanes behind the
// segment's end simply load the following entries -- in bounds or zeroed by the descriptor's range check -- and are masked at
// the atomic), then v_cvt_f32_f16, v_mul_f32 (weight x rating: 24 x 11 significant bits, rounded once to fp32 -- 6e-8
// relative per contribution, the same error the fp32 weight already carries), v_cvt_f64_f32, the LDS address and the
// lane < length compare.
// a group of 8 slice loads of the packed walk, in flight or landed
struct PkGroup {
    int L[FY_COOC_NB];
    float W[FY_COOC_NB];
    uint32_t pk[FY_COOC_NB];
};
// issues the 8 slice loads of the descriptor slots [g, g + 8) of a batch (d, w)
__device__ __forceinline__ void cooc_pk_issue(const __amdgpu_buffer_rsrc_t rsrc, int lane4, PkGroup& G, const int2& d, float w, int g) {
#pragma unroll
    for (int q = 0; q < FY_COOC_NB; q++) {
        const int F = __builtin_amdgcn_readlane(d.x, g + q);
        G.L[q] = __builtin_amdgcn_readlane(d.y, g + q);
        G.W[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w), g + q));
        G.pk[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane4, F * 4, 0);
    }
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cooc_pk_rsrc(const CoocArgs& A) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(A.csr_pk), 0, (int)A.pk_bytes, 0x00020000);
}
template <class ACC>
__device__ __forceinline__ void cooc_accumulate_pk(const CoocArgs& A, ACC* __restrict__ acc, int s_begin, int s_end, SegBatch first,
                                                   int rmask = -1 /* symmetric walk, own chunk: only columns > rmask (chunk-relative) count */) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = cooc_pk_rsrc(A);
    const int lane4 = lane * 4;
    constexpr int NB = FY_COOC_NB;
    using Group = PkGroup;
    auto issue = [&](Group& G, const int2& d, float w, int g) __attribute__((always_inline)) { cooc_pk_issue(rsrc, lane4, G, d, w, g); };
    const double fx = A.fx_scale;
    // accumulator of chunk-relative column c (cooc_acc_index) as ONE 24-bit multiply-add without a select:
    // planar (c & 3) * quarter + (c >> 2), linear (c & 0) * 0 + (c >> 0)
    const uint32_t ix_mask = A.acc_quarter ? 3u : 0u, ix_shift = A.acc_quarter ? 2u : 0u, ix_quarter = (uint32_t)A.acc_quarter;
    auto commit = [&](Group& G) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const float x = __half2float(__ushort_as_half((unsigned short)(G.pk[q] >> 16)));
            if constexpr (std::is_same<ACC, uint32_t>::value) {
                // 32-bit fixed point (item similarity on ratings that are multiples of 2^-m, fx_scale = 2^2m): every product is an
                // integer below 2^24 -- exact in fp32 -- and every sum stays below 2^32 (fy_rm2.hip: gram_half_build): ds_add_u32
                const uint32_t v = (uint32_t)(G.W[q] * x * (float)fx);
                if (lane < G.L[q] && (int)(G.pk[q] & 0xFFFFu) > rmask) atomicAdd(&acc[cooc_acc_index((int)(G.pk[q] & 0xFFFFu), A.acc_quarter)], v);   // ds_add_u32
            } else if constexpr (std::is_same<ACC, unsigned long long>::value) {
                // round(p * 2^k) without a 64-bit conversion: p * 2^k + 2^52 has the integer in its mantissa (0 <= p * 2^k < 2^52)
                const double d = fma((double)(G.W[q] * x), fx, 4503599627370496.0);
                unsigned long long v = (unsigned long long)__double_as_longlong(d) & 0xFFFFFFFFFFFFFull;
                // (round 3, from the ISA: the index was a 64-bit multiply-add plus a select between the planar and the linear layout,
                // and value and index were computed INSIDE an exec-masked branch per segment -- s_and_saveexec, s_cbranch_execz.  Now a
                // 24-bit multiply-add, and everything but the atomic in front of the mask: the masked block is one instruction, no branch.)
                const uint32_t c = G.pk[q] & 0xFFFFu;
                uint32_t at = __umul24(c & ix_mask, ix_quarter) + (c >> ix_shift);
                asm volatile("" : "+v"(at), "+v"(v));
                if (lane < G.L[q] && (int)c > rmask) atomicAdd(&acc[at], v);   // ds_add_u64
            } else {
                const ACC v = (ACC)(G.W[q] * x);
                if (lane < G.L[q] && (int)(G.pk[q] & 0xFFFFu) > rmask) atomicAdd(&acc[cooc_acc_index((int)(G.pk[q] & 0xFFFFu), A.acc_quarter)], v);   // ds_add_f64
            }
        }
    };
    int batch = 0;
    if (cooc_seg_slot(s_begin, 0, wave, nwaves, 0) >= s_end) return;
    int2 d = first.d;
    float w = first.w;
    while (true) {   // wave-uniform
        const int nloc = __popcll(__ballot(cooc_seg_slot(s_begin, batch, wave, nwaves, lane) < s_end));
        const int s_next = cooc_seg_slot(s_begin, batch + 1, wave, nwaves, lane);
        int2 dn = make_int2(0, 0);
        float wn = 0.0f;
        if (s_next < s_end) { dn = A.seg[s_next]; wn = A.seg_w[s_next]; }
        Group GA, GB;      // two groups of NB loads in flight; straight-line body (see cooc_accumulate_segments)
        issue(GA, d, w, 0);
#pragma unroll 1
        for (int g = 0; g < 64; g += 2 * NB) {
            issue(GB, d, w, g + NB);
            commit(GA);
            if (g + 2 * NB < 64) issue(GA, d, w, g + 2 * NB);
            commit(GB);
            if (g + 2 * NB >= nloc) break;
        }
        d = dn;
        w = wn;
        batch++;
        if (cooc_seg_slot(s_begin, batch, wave, nwaves, 0) >= s_end) break;
    }
}

// Accumulates chunk `ch` of row `row` into the workgroup's dynamic LDS (fy_cooc_acc[0..CH) as fp64, zeroed by the caller;
// all threads call this).
//
// Unit of work = one SEGMENT (<= 64 entries of one rater's row slice, one entry per lane).  The segments of an item row
// are contiguous in the segment table, so a wave simply takes 64 of them at a time (wave w: segments 64 w, 64 (w + nwaves),
// ...), broadcasts their descriptors with v_readlane, eight at a time: eight coalesced loads, then eight LDS atomics.
// Heavy users -- raters of very many rows with slices of thousands of entries, half of all rater visits at ML-25M shape --
// are spread over all waves of the workgroup by construction; there is no per-rater loop left.
// History (rocprof, ML-25M shape, ms per M build): one scalar chain per rater 110; per-lane binary-search expansion 362;
// 4 raters per wave-step 84; + pipelined long-slice loop 48 (of which 28 were waves idling behind the wave that held a
// long slice); LDS queue for the remainders 104; segment s of rater r -> wave (r + s) mod 16 with every wave reading all
// metadata 90; persistent workgroups 47; segments precomputed per CSC entry: DESIGN.md section 7.
template <bool PK = false>
__device__ __forceinline__ void cooc_accumulate_row(const CoocArgs& A, int row, int ch, int lrow = 0) {
    int s_begin, s_end;
    if (A.item_seg) {
        const int2 se = A.item_seg[(int64_t)lrow * A.nch + ch];
        s_begin = se.x;
        s_end = se.y;
    } else {
        const int pair = A.rank_pair[A.pbase + row];
        int e0 = A.pair_start[pair], e1 = A.pair_start[pair + 1];
        if (A.local_start) { e0 = A.local_start[lrow]; e1 = A.local_start[lrow + 1]; }
        const int32_t* __restrict__ sp = A.seg_ptr + (int64_t)ch * (A.nq + 1) - A.q0;
        s_begin = sp[e0];
        s_end = sp[e1];
    }
    if constexpr (PK) cooc_accumulate_pk<double>(A, fy_cooc_acc, s_begin, s_end, cooc_first_batch(A, s_begin, s_end));
    else cooc_accumulate_segments<false, double>(A, fy_cooc_acc, s_begin, s_end, ch * A.CH, cooc_first_batch(A, s_begin, s_end));
}

// segment range of every (row, chunk) item of a launch (CoocArgs::item_seg)
__attribute__((unused)) static __global__ void k_item_segments(CoocArgs A, int2* __restrict__ out) {
    const int n = A.nrows * A.nch;
    const int stride = A.row_stride ? A.row_stride : 1;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int lrow = t / A.nch, ch = t % A.nch;
        const int pair = A.rank_pair[A.pbase + A.row0 + lrow * stride];
        int e0 = A.pair_start[pair], e1 = A.pair_start[pair + 1];
        if (A.local_start) { e0 = A.local_start[lrow]; e1 = A.local_start[lrow + 1]; }
        const int32_t* __restrict__ sp = A.seg_ptr + (int64_t)ch * (A.nq + 1) - A.q0;
        out[t] = make_int2(sp[e0], sp[e1]);
    }
}

// item list of the RM2 row kernel (CoocArgs::item_id / item_seg): rows ascending, inside a row chunks ascending.  Full walk:
// t = row * nch + ch.  Half walk: the rows of chunk c have nch - c items (chunks c .. nch - 1), so
// t = sum_{c' < c} rows(c') * (nch - c') + (row - c * CH) * (nch - c) + (ch - c), rows(c') = CH (the last chunk may be short,
// but nothing follows it).  One thread per (row, chunk) pair; pairs in front of the row's own chunk are skipped.
__host__ __device__ inline int64_t cooc_half_item_index(int32_t row, int32_t ch, int32_t CH, int32_t nch) {
    const int64_t c = row / CH;
    // sum_{c' < c} CH * (nch - c') = CH * (c * nch - c (c - 1) / 2)
    return (int64_t)CH * (c * nch - c * (c - 1) / 2) + (int64_t)(row - c * CH) * (nch - c) + (ch - c);
}
inline int64_t cooc_item_count(int32_t nrows, int32_t CH, int32_t nch, bool half) {
    if (!half) return (int64_t)nrows * nch;
    return nrows > 0 ? cooc_half_item_index(nrows - 1, nch - 1, CH, nch) + 1 : 0;
}
__device__ __forceinline__ void item_list_body(const CoocArgs& A, int2* __restrict__ seg_out, int32_t* __restrict__ id_out) {
    const int n = A.nrows * A.nch;
    const int stride = A.row_stride ? A.row_stride : 1;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int lrow = t / A.nch, ch = t % A.nch;
        const int row = A.row0 + lrow * stride;
        int64_t at = t;
        if (A.half && A.half_rows > 0) {            // symmetric panel mode (row0 = 0, stride 1): head rows cut, tail rows whole, both over the head chunks
            if (row < A.half_rows) {
                if (ch < row / A.CH || ch >= A.head_chunks) continue;
                at = cooc_half_item_index(row, ch, A.CH, A.head_chunks);
            } else {
                if (ch >= A.tail_chunks) continue;
                at = cooc_half_item_index(A.half_rows - 1, A.head_chunks - 1, A.CH, A.head_chunks) + 1 + (int64_t)(row - A.half_rows) * A.tail_chunks + ch;
            }
        } else if (A.half) {
            if (ch < row / A.CH) continue;
            at = cooc_half_item_index(row, ch, A.CH, A.nch);
        } else if (A.tail_chunks > 0 && lrow >= A.tail_row0) {
            if (ch >= A.tail_chunks) continue;
            at = (int64_t)A.tail_row0 * A.nch + (int64_t)(lrow - A.tail_row0) * A.tail_chunks + ch;
        }
        const int pair = A.rank_pair[A.pbase + row];
        int e0 = A.pair_start[pair], e1 = A.pair_start[pair + 1];
        if (A.local_start) { e0 = A.local_start[lrow]; e1 = A.local_start[lrow + 1]; }
        const int32_t* __restrict__ sp = A.seg_ptr + (int64_t)ch * (A.nq + 1) - A.q0;
        seg_out[at] = make_int2(sp[e0], sp[e1]);
        id_out[at] = (lrow << 8) | ch;
    }
}
__attribute__((unused)) static __global__ void k_item_list(CoocArgs A, int2* __restrict__ seg_out, int32_t* __restrict__ id_out) {
    item_list_body(A, seg_out, id_out);
}

// segment table of one cluster from its chunk_off table; returns the number of segments (synchronises once)
struct SegTable {
    int64_t n_seg = 0;     // segments in the table (an upper bound when the table was sized without a round trip)
    DevBuf<int32_t> cnt, scratch;   // build_segments' temporaries: they live as long as the table (a lane may still be reading them)
    DevBuf<int32_t> ptr;   // nch * (nq + 1)
    DevBuf<int2> seg;
    DevBuf<float> w;
    // a table that is a RANGE of the job's one table over all clusters (fy_rm2.hip: build_tables_all) owns nothing: views
    const int32_t* v_ptr = nullptr;
    const int2* v_seg = nullptr;
    const float* v_w = nullptr;
    const int32_t* ptr_() const { return v_ptr ? v_ptr : ptr.get(); }
    const int2* seg_() const { return v_seg ? v_seg : seg.get(); }
    const float* w_() const { return v_w ? v_w : w.get(); }
};
// half_row_of_entry != nullptr: symmetric walk (only the columns behind the entry's own row; fy_rm2.hip, struct Half)
// only_rows_of_entry != nullptr: CSC entries of the rows in front of only_rows_from get no segments (tail-row launches)
// max_segments > 0: an upper bound of the number of segments -- the table is allocated for it and the call does not wait for
// the device (otherwise it reads the exact count back: two host round trips)
void build_segments(Context* ctx, const int32_t* csc_slot, const float* csc_w, const int32_t* chunk_off, int32_t slot_base,
                    int32_t q0, int32_t nq, int32_t nch, SegTable& out, hipStream_t st = nullptr,
                    const int32_t* half_row_of_entry = nullptr, const int32_t* csr_idx = nullptr, int32_t CH = 0,
                    const int32_t* only_rows_of_entry = nullptr, int32_t only_rows_from = 0, int64_t max_segments = 0,
                    const int32_t* samples = nullptr);
// every 64th entry of csr_idx: what the symmetric cut of the segment tables searches instead of the rows themselves
void build_csr_samples(Context* ctx, int64_t nnz, const int32_t* csr_idx, DevBuf<int32_t>& samp, hipStream_t st = nullptr);

// chunk_off table for one cluster: one thread per (slot, boundary)
void build_chunk_offsets(Context* ctx, const int32_t* rowptr, const int32_t* csr_idx, int32_t slot_base, int32_t n_slots,
                         int32_t CH, int32_t nch, int32_t* chunk_off, hipStream_t st = nullptr);

// columns of LDS accumulators a workgroup of the RM2 row kernel allocates for chunks of CH columns: its epilogue reads whole
// 256-column blocks up to the padded row length, which overshoots a chunk whose width is not a multiple of 256 by < 256
__host__ __device__ inline int cooc_lds_columns(int CH) { return (CH % 256 == 0) ? CH : ((CH + 255) / 256) * 256 + 256; }

// picks the chunk width for a cluster with Ic items: whole row when it fits the LDS budget.  Chunks are multiples of 256
// columns whenever there are several (a wave of the epilogue then owns exactly one 256-column block, fy_rm2.hip).
inline void pick_chunks(int32_t Ic, int32_t max_ch, int32_t& CH, int32_t& nch) {
    if (Ic <= max_ch) {
        CH = (int32_t)round_up(Ic > 0 ? Ic : 1, 64);
        nch = 1;
    } else {
        const int32_t cap = max_ch >= 256 ? (max_ch / 256) * 256 : max_ch;
        const int32_t gran = max_ch >= 256 ? 256 : 64;
        nch = (int32_t)ceil_div(Ic, cap);
        CH = (int32_t)std::min<int64_t>(cap, round_up(ceil_div(Ic, nch), gran));
        nch = (int32_t)ceil_div(Ic, CH);
    }
}

}  // namespace fy
