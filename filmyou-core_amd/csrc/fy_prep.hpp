// fy_prep.hpp -- the rating matrix as it lives in HBM: cluster-major CSR (user rows) + CSC (item columns).
//
// Layout (all int32 offsets; nnz < 2^31):
//   users   : dense index du = rank of the raw user id (ascending).  "slot" = position in cluster-major order,
//             inside a cluster by descending degree (heavy rows first => balanced tail).  ucstart[c] .. ucstart[c+1]
//             are the slots of cluster c.
//   items   : per cluster only the items rated by somebody of that cluster exist (AbstractRM2Reducer.java:238-272).
//             A (cluster, item) "pair" gets the compact index idx = popularity rank inside the cluster (0 = most
//             rated; ties ascending raw id), so the hot rows of the co-rating matrix are the low rows.
//             pcstart[c] + idx addresses per-cluster item arrays ("rank order").
//   CSR     : rowptr[slot], csr_idx (compact idx, ascending inside a row), csr_r (raw rating), csr_x, csr_e.
//   CSC     : pair order = (cluster, raw item) ascending; pair_start[pair] .. pair_start[pair+1] index csc_slot /
//             csc_r / csc_x; inside a column ascending raw user id.
#pragma once
#include "fy_common.hpp"

namespace fy {

struct Prepared {
    Context* ctx = nullptr;
    int64_t nnz = 0;          // ratings kept (score > 0)
    int32_t nU = 0, nI = 0, nP = 0, K = 0;
    int64_t sum_deg2 = 0;     // sum_u n_u^2
    std::vector<int64_t> cluster_deg2;   // [c]: sum of n_u^2 over the users of cluster c (host; upper bounds of the segment tables)
    bool ratings_fp16_exact = false;   // every kept rating is exactly representable in fp16 (enables the packed CSR of fy_cooc.hpp)
    int ratings_frac_bits = 0;         // every kept rating is a multiple of 2^-this (half stars: 1; 9 = finer than 2^-8)
    bool ratings_positive = false;     // every kept rating is > 0 (always in RM2; the item-similarity job keeps ratings <= 0): the
                                       // fixed-point accumulators of the row kernel add unsigned contributions
    // users, dense (ascending raw id)
    DevBuf<int32_t> uid, ucluster, udeg, slot2du, du2slot;
    DevBuf<double> usum;
    // items, dense (ascending raw id)
    DevBuf<int32_t> iid;
    // cluster tables (host + device)
    std::vector<int32_t> csize, ucstart, pcstart;   // K, K+1, K+1
    std::vector<int32_t> cluster_q;                  // K+1: first CSC entry of every cluster
    DevBuf<int32_t> d_ucstart, d_pcstart, d_csize;
    // pairs, (cluster, item) ascending
    DevBuf<int32_t> pair_cluster, pair_di, pair_start /* nP+1 */, pair_rank, rank_pair;
    // per-cluster item arrays in rank order (index pcstart[c] + idx)
    DevBuf<int32_t> rank_item_raw;
    // CSC (pair order)
    DevBuf<int32_t> csc_slot, csc_pair;
    DevBuf<float> csc_r;
    // CSR (slot order)
    DevBuf<int32_t> rowptr /* nU+1 */, csr_idx;
    DevBuf<float> csr_r;
    // per-slot work model n_u * (I_c - n_u), inclusive prefix (host copy, used to cut rank ranges)
    std::vector<int64_t> work_prefix;
};

// Builds every structure that does not depend on the job's statistics.  map_* are host arrays (may be empty).
// Throws fy::Failure.
void build_structure(Context* ctx, const fy_ratings* R, int32_t n_clusters, int64_t n_map, const int32_t* map_user,
                     const int32_t* map_cluster, const int32_t* cluster_count, bool keep_nonpositive, Prepared& P,
                     const int32_t* cl_of_raw = nullptr /* device table raw user id -> cluster (max_user + 1 entries), when the caller has one */);

// Sharded prep (several ranks, at least as many non-empty clusters as ranks): decides which rank owns which WHOLE cluster from counts
// every rank finds identically in the replicated COO (owner[c] = rank, `world` for an empty cluster) and copies the kept ratings of this
// rank's clusters into `mine` (input order kept).  false = this job does not shard (fewer clusters than ranks, ids too sparse): the
// caller preps the replicated ratings as before.  Failures that every rank sees are thrown (fy::Failure) by every rank.
bool shard_ratings_by_cluster(Context* ctx, const fy_ratings* R, int32_t n_clusters, int64_t n_map, const int32_t* map_user,
                              const int32_t* map_cluster, int rank, int world, fy_ratings& mine, std::vector<int32_t>& owner,
                              DevBuf<int32_t>& cl_of_raw /* out: the device table raw user id -> cluster it built */);

// fills R->max_user / R->max_item (one pass over the COO; called by fy_ratings_create)
void ratings_id_bounds(Context* ctx, fy_ratings* R);

// slot range [lo, hi) of `rank` out of `world`, cut on the work prefix (identical on every rank)
void rank_slot_range(const Prepared& P, int rank, int world, int32_t& lo, int32_t& hi);

// generic device helpers implemented with rocPRIM (radix sort / scan), all on ctx->stream
void sort_pairs_u64_u32(Context*, uint64_t* kin, uint64_t* kout, uint32_t* vin, uint32_t* vout, size_t n, int end_bit = 64);
void sort_pairs_u64_f32(Context*, uint64_t* kin, uint64_t* kout, float* vin, float* vout, size_t n, int end_bit = 64, int begin_bit = 0);
void sort_keys_u64(Context*, uint64_t* kin, uint64_t* kout, size_t n, int end_bit = 64, int begin_bit = 0);
void sort_pairs_u64_u64(Context*, uint64_t* kin, uint64_t* kout, uint64_t* vin, uint64_t* vout, size_t n, int end_bit = 64);
void inclusive_scan_u32(Context*, const uint32_t* in, uint32_t* out, size_t n);
// scratch: a buffer of the caller for the scan's temporary storage (grown when too small); it must outlive the work queued on st
void exclusive_scan_i32(Context*, const int32_t* in, int32_t* out, size_t n, hipStream_t st = nullptr, DevBuf<char>* scratch = nullptr);
void inclusive_scan_i64(Context*, const int64_t* in, int64_t* out, size_t n);
// out[t] = src[index[t]] for a host list of positions (one gather launch, one copy back; synchronises the context)
void gather_to_host_i32(Context*, const int32_t* src, const std::vector<int32_t>& index, int32_t* out);
void gather_to_host_i64(Context*, const int64_t* src, const std::vector<int32_t>& index, int64_t* out);

}  // namespace fy
