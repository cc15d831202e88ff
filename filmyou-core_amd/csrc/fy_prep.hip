// fy_prep.hip -- COO ratings -> cluster-major CSR + CSC in HBM (the reference's map / shuffle / group phases).
//
// What each step replaces in the reference (M/ = src/main/java/es/udc/fi/dc/irlab/):
//   score > 0 filter          M/rm/SimpleScoreByUserHDFSMapper.java:37-40, ScoreByClusterHDFSMapper.java:40-41
//   user -> cluster routing   M/common/AbstractByClusterMapper.java:46-79 (missing user -> 0)
//   cluster sizes             M/nmf/clustering/CountClustersJob (the clusteringCount side file), validated here
//   group ratings by cluster  the RM2-3 shuffle + secondary sort (M/rm/RM2Job.java:225-258)
//   per-cluster item set      M/rm/AbstractRM2Reducer.java:238-272 (createUserAndItemMappings)
// Sorting and scanning are plumbing and use rocPRIM; every other step is a hand-written kernel.
#include <cstring>
#include <cstdlib>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <numeric>

#include <hip/hip_fp16.h>

#include "fy_prep.hpp"

namespace fy {

// ---------------------------------------------------------------- rocPRIM wrappers
template <class K, class V>
static void sort_pairs_impl(Context* c, K* kin, K* kout, V* vin, V* vout, size_t n, int end_bit, int begin_bit = 0) {
    if (n == 0) return;
    size_t tmp = 0;
    FY_HIP(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, begin_bit, end_bit, c->stream));
    DevBuf<char> t(c, tmp);
    FY_HIP(rocprim::radix_sort_pairs(t.get(), tmp, kin, kout, vin, vout, n, begin_bit, end_bit, c->stream));
}
void sort_pairs_u64_u32(Context* c, uint64_t* kin, uint64_t* kout, uint32_t* vin, uint32_t* vout, size_t n, int end_bit) {
    sort_pairs_impl(c, kin, kout, vin, vout, n, end_bit);
}
void sort_pairs_u64_f32(Context* c, uint64_t* kin, uint64_t* kout, float* vin, float* vout, size_t n, int end_bit, int begin_bit) {
    sort_pairs_impl(c, kin, kout, vin, vout, n, end_bit, begin_bit);
}
void sort_keys_u64(Context* c, uint64_t* kin, uint64_t* kout, size_t n, int end_bit, int begin_bit) {
    if (n == 0) return;
    size_t bytes = 0;
    FY_HIP(rocprim::radix_sort_keys(nullptr, bytes, kin, kout, n, (unsigned)begin_bit, (unsigned)end_bit, c->stream));
    DevBuf<char> tmp(c, bytes);
    FY_HIP(rocprim::radix_sort_keys(tmp.get(), bytes, kin, kout, n, (unsigned)begin_bit, (unsigned)end_bit, c->stream));
}
void sort_pairs_u64_u64(Context* c, uint64_t* kin, uint64_t* kout, uint64_t* vin, uint64_t* vout, size_t n, int end_bit) {
    sort_pairs_impl(c, kin, kout, vin, vout, n, end_bit);
}
void inclusive_scan_u32(Context* c, const uint32_t* in, uint32_t* out, size_t n) {
    if (n == 0) return;
    size_t tmp = 0;
    FY_HIP(rocprim::inclusive_scan(nullptr, tmp, in, out, n, rocprim::plus<uint32_t>(), c->stream));
    DevBuf<char> t(c, tmp);
    FY_HIP(rocprim::inclusive_scan(t.get(), tmp, in, out, n, rocprim::plus<uint32_t>(), c->stream));
}
void exclusive_scan_i32(Context* c, const int32_t* in, int32_t* out, size_t n, hipStream_t st, DevBuf<char>* scratch) {
    if (n == 0) return;
    if (!st) st = c->stream;
    size_t tmp = 0;
    FY_HIP(rocprim::exclusive_scan(nullptr, tmp, in, out, int32_t(0), n, rocprim::plus<int32_t>(), st));
    if (scratch) {
        // the caller's buffer outlives everything it queues on the lane (hipMallocAsync / hipFreeAsync around every scan of a lane cost
        // the HOST ~1 ms per call: 48 of the 82 ms of a 50-cluster job were spent queueing the clusters' fronts)
        if (scratch->size() < tmp || !scratch->get()) scratch->alloc(c, std::max<size_t>(tmp, 4096));
        FY_HIP(rocprim::exclusive_scan(scratch->get(), tmp, in, out, int32_t(0), n, rocprim::plus<int32_t>(), st));
    } else if (st == c->stream) {
        DevBuf<char> t(c, tmp);
        FY_HIP(rocprim::exclusive_scan(t.get(), tmp, in, out, int32_t(0), n, rocprim::plus<int32_t>(), st));
    } else {
        // on a lane the temporary must not go back to the (single-stream) allocator while the lane still uses it
        void* t = nullptr;
        FY_HIP(hipMallocAsync(&t, tmp ? tmp : 1, st));
        FY_HIP(rocprim::exclusive_scan(t, tmp, in, out, int32_t(0), n, rocprim::plus<int32_t>(), st));
        FY_HIP(hipFreeAsync(t, st));
    }
}
void inclusive_scan_i64(Context* c, const int64_t* in, int64_t* out, size_t n) {
    if (n == 0) return;
    size_t tmp = 0;
    FY_HIP(rocprim::inclusive_scan(nullptr, tmp, in, out, n, rocprim::plus<int64_t>(), c->stream));
    DevBuf<char> t(c, tmp);
    FY_HIP(rocprim::inclusive_scan(t.get(), tmp, in, out, n, rocprim::plus<int64_t>(), c->stream));
}

// ---------------------------------------------------------------- kernels
enum { ERR_NEG_ID = 1, ERR_DUP = 2, ERR_CLUSTER_RANGE = 4, NOTE_NOT_FP16 = 8, NOTE_NONPOSITIVE = 16 };

static inline int grid_for(int64_t n, int block = 256, int cap = 256 * 16) {
    int64_t g = ceil_div(n, block);
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

// largest user / item id among the kept ratings: the sort keys are packed into as few bits as the ids need (every 8 bits
// less is one radix pass over 25 M pairs less)
__global__ void k_max_ids(int64_t n, const int32_t* __restrict__ user, const int32_t* __restrict__ item,
                          const float* __restrict__ score, int keep_nonpositive, int32_t* __restrict__ max_ids) {
    int32_t mu = -1, mi = -1;
    bool not_half = false;      // max_ids[2] = 1: some score (NaN aside: no job keeps one) is not exactly representable in fp16
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const float s = score[t];
        if (keep_nonpositive == 2 || (keep_nonpositive ? (s == s) : (s > 0.0f))) { mu = max(mu, user[t]); mi = max(mi, item[t]); }   // 2: every entry
        if (s == s && __half2float(__float2half(s)) != s) not_half = true;
    }
    if (__ballot(not_half) && (threadIdx.x & 63) == 0) max_ids[2] = 1;
    for (int o = 32; o > 0; o >>= 1) {
        mu = max(mu, __shfl_down(mu, o, 64));
        mi = max(mi, __shfl_down(mi, o, 64));
    }
    // one pair of atomics per workgroup (thousands of waves on one address serialise in the L2: ~5 ns each)
    __shared__ int32_t sh[2][16];
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = mu; sh[1][threadIdx.x >> 6] = mi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) { mu = max(mu, sh[0][w]); mi = max(mi, sh[1][w]); }
        if (mu >= 0) atomicMax(&max_ids[0], mu);
        if (mi >= 0) atomicMax(&max_ids[1], mi);
    }
}

// key = (user << ib) | item for kept ratings; dropped ones get the key of user `drop_user` (beyond every kept user: they
// sort to the end)
__global__ void k_user_item_keys(int64_t n, const int32_t* __restrict__ user, const int32_t* __restrict__ item,
                                 const float* __restrict__ score, int keep_nonpositive, int ib, uint32_t drop_user,
                                 uint64_t* __restrict__ keys, unsigned long long* __restrict__ kept, int* __restrict__ err, int pb) {
    // pb = 16 (packed mode: every score is exactly a half): the key carries the rating in its low 16 bits and the sorts move keys alone
    unsigned long long local = 0;
    bool not_half = false, not_pos = false;
    unsigned frac = 0;          // bit 8 + m: some kept rating needs m fractional bits (m = 9: more than eight)
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const float s = score[t];
        const bool keep = keep_nonpositive ? (s == s) : (s > 0.0f);   // NaN never passes "score > 0"
        uint64_t k = (uint64_t)drop_user << (ib + pb);
        if (keep) {
            const int32_t u = user[t], i = item[t];
            if (u < 0 || i < 0) atomicOr(err, ERR_NEG_ID);
            if (__half2float(__float2half(s)) != s) not_half = true;
            if (!(s > 0.0f)) not_pos = true;
            {
                float t = fabsf(s);
                int m = 0;
                while (m < 9 && t != floorf(t)) { t *= 2.0f; m++; }
                frac |= 1u << (8 + m);
            }
            k = ((uint64_t)(uint32_t)u << (ib + pb)) | ((uint64_t)(uint32_t)i << pb) | (pb ? (uint64_t)__half_as_ushort(__float2half(s)) : 0ull);
            local++;
        }
        keys[t] = k;
    }
    // wave reduce, workgroup reduce, then ONE atomic per counter and workgroup (round 4: an atomic per wave -- 16 384 waves x 4 atomics
    // on two addresses -- was what this kernel took its 0.25 ms for, whatever the number of ratings)
    if (not_half) frac |= NOTE_NOT_FP16;
    if (not_pos) frac |= NOTE_NONPOSITIVE;
    for (int o = 32; o > 0; o >>= 1) {
        local += __shfl_down(local, o, 64);
        frac |= (unsigned)__shfl_down((int)frac, o, 64);
    }
    __shared__ unsigned long long sh_local[16];
    __shared__ unsigned sh_frac[16];
    if ((threadIdx.x & 63) == 0) { sh_local[threadIdx.x >> 6] = local; sh_frac[threadIdx.x >> 6] = frac; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) { local += sh_local[w]; frac |= sh_frac[w]; }
        if (local) atomicAdd(kept, local);
        if (frac) atomicOr(err, (int)frac);
    }
}

__global__ void k_heads_hi32(int64_t n, const uint64_t* __restrict__ keys, uint32_t* __restrict__ head, int check_dup,
                             int* __restrict__ err, int ib) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = 1;
        if (t > 0) {
            const uint64_t a = keys[t - 1], b = keys[t];
            h = (a >> ib) != (b >> ib);
            if (check_dup && a == b) atomicOr(err, ERR_DUP);
        }
        head[t] = h;
    }
}

// heads of the (cluster, item) columns; two neighbours of one column with the same slot are one user's two ratings of one item
__global__ void k_heads_full_dup(int64_t n, const uint64_t* __restrict__ keys, const uint64_t* __restrict__ vals, uint32_t* __restrict__ head, int* __restrict__ err) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const bool same = t > 0 && keys[t] == keys[t - 1];
        head[t] = !same;
        if (same && (vals[t] >> 32) == (vals[t - 1] >> 32)) atomicOr(err, ERR_DUP);
    }
}
__global__ void k_heads_full(int64_t n, const uint64_t* __restrict__ keys, uint32_t* __restrict__ head) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        head[t] = (t == 0) || (keys[t] != keys[t - 1]);
}

// at every user head: uid[du] = raw id, ustart[du] = t
__global__ void k_scatter_users(int64_t n, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ head,
                                const uint32_t* __restrict__ du1, int32_t* __restrict__ uid, int32_t* __restrict__ ustart, int ib) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        if (head[t]) {
            const uint32_t d = du1[t] - 1;
            uid[d] = (int32_t)(keys[t] >> ib);
            ustart[d] = (int32_t)t;
        }
}

// one wave per user: s_u = sum of (double) score in a fixed order (DoubleSumAndCountReducer.java:35-38), degree
__device__ __forceinline__ float fy_key_rating(uint64_t key) { return __half2float(__ushort_as_half((unsigned short)(key & 0xFFFFu))); }
__global__ void k_user_sums(int32_t nU, const int32_t* __restrict__ ustart, const float* __restrict__ score, const uint64_t* __restrict__ pkeys,
                            double* __restrict__ usum, int32_t* __restrict__ udeg) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int32_t u = blockIdx.x * wpb + (threadIdx.x >> 6); u < nU; u += gridDim.x * wpb) {
        const int32_t a = ustart[u], b = ustart[u + 1];
        double s = 0.0;
        if (pkeys) { for (int32_t t = a + lane; t < b; t += 64) s += (double)fy_key_rating(pkeys[t]); }      // (packed mode)
        else for (int32_t t = a + lane; t < b; t += 64) s += (double)score[t];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) { usum[u] = s; udeg[u] = b - a; }
    }
}

// cluster of a user: last entry of the (stably sorted) clustering map with that user id, 0 when absent (Q2)
__global__ void k_lookup_cluster(int32_t nU, const int32_t* __restrict__ uid, int64_t n_map,
                                 const uint64_t* __restrict__ map_sorted /* user : position in the file */, const uint32_t* __restrict__ map_cluster_sorted,
                                 int32_t K, int32_t* __restrict__ ucluster, int* __restrict__ err) {
    for (int32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < nU; u += gridDim.x * blockDim.x) {
        const uint32_t id = (uint32_t)uid[u];
        int64_t lo = 0, hi = n_map;            // upper bound of id in the hi32 part
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((uint32_t)(map_sorted[mid] >> 32) <= id) lo = mid + 1; else hi = mid;
        }
        int32_t c = 0;
        if (lo > 0 && (uint32_t)(map_sorted[lo - 1] >> 32) == id) c = (int32_t)map_cluster_sorted[lo - 1];
        if (c < 0 || c >= K) { atomicOr(err, ERR_CLUSTER_RANGE); c = 0; }
        ucluster[u] = c;
    }
}

// the same from a table indexed by raw user id (dense ids: no sort of the map, no search)
__global__ void k_lookup_cluster_table(int32_t nU, const int32_t* __restrict__ uid, const int32_t* __restrict__ cl_of_raw, int32_t K,
                                       int32_t* __restrict__ ucluster, int* __restrict__ err) {
    for (int32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < nU; u += gridDim.x * blockDim.x) {
        int32_t c = cl_of_raw[uid[u]];
        if (c < 0 || c >= K) { atomicOr(err, ERR_CLUSTER_RANGE); c = 0; }
        ucluster[u] = c;
    }
}

__global__ void k_slot_keys(int32_t nU, const int32_t* __restrict__ ucluster, const int32_t* __restrict__ udeg,
                            uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int32_t* __restrict__ csize) {
    for (int32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < nU; u += gridDim.x * blockDim.x) {
        keys[u] = ((uint64_t)(uint32_t)ucluster[u] << 32) | (0xFFFFFFFFu - (uint32_t)udeg[u]);
        vals[u] = (uint32_t)u;
        // one atomic per wave when the whole wave sits in one cluster (always, with a single cluster)
        const int c = ucluster[u];
        const int c0 = __shfl(c, __ffsll((long long)__ballot(1)) - 1, 64);
        const unsigned long long same = __ballot(c == c0);
        if (same == __ballot(1)) {
            if ((threadIdx.x & 63) == __ffsll((long long)same) - 1) atomicAdd(&csize[c0], (int)__popcll(same));
        } else {
            atomicAdd(&csize[c], 1);
        }
    }
}
// the same with the cluster sizes counted in LDS first (K <= SLOT_KEYS_LDS clusters: 50 clusters took one global atomic per user
// on 50 addresses -- 86 us for the 20 000 users of a rank's share)
constexpr int SLOT_KEYS_LDS = 4096;
__global__ void k_slot_keys_lds(int32_t nU, int32_t K, const int32_t* __restrict__ ucluster, const int32_t* __restrict__ udeg,
                                uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int32_t* __restrict__ csize) {
    __shared__ int32_t sh[SLOT_KEYS_LDS];
    for (int c = threadIdx.x; c < K; c += blockDim.x) sh[c] = 0;
    __syncthreads();
    for (int32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < nU; u += gridDim.x * blockDim.x) {
        const int c = ucluster[u];
        keys[u] = ((uint64_t)(uint32_t)c << 32) | (0xFFFFFFFFu - (uint32_t)udeg[u]);
        vals[u] = (uint32_t)u;
        atomicAdd(&sh[c], 1);       // (k_lookup_cluster has put every user into [0, K))
    }
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += blockDim.x)
        if (sh[c]) atomicAdd(&csize[c], sh[c]);
}

__global__ void k_invert_perm(int32_t n, const uint32_t* __restrict__ perm, int32_t* __restrict__ fwd, int32_t* __restrict__ inv) {
    for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        fwd[s] = (int32_t)perm[s];
        inv[perm[s]] = s;
    }
}

// (cluster : raw item) key of every rating; the payload travels with it through the sort: (slot of the rater, rating bits),
// so the CSC arrays fall out of the sorted values without a gather
__global__ void k_cluster_item_keys(int64_t n, const uint64_t* __restrict__ ukeys, const uint32_t* __restrict__ du1,
                                    const int32_t* __restrict__ ucluster, const int32_t* __restrict__ du2slot,
                                    const float* __restrict__ score_um, uint64_t* __restrict__ keys, uint64_t* __restrict__ vals, int ib) {
    const uint64_t mask = ((uint64_t)1 << ib) - 1;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t du = du1[t] - 1;
        keys[t] = ((uint64_t)(uint32_t)ucluster[du] << ib) | (ukeys[t] & mask);
        vals[t] = ((uint64_t)(uint32_t)du2slot[du] << 32) | __float_as_uint(score_um[t]);
    }
}

// packed mode: ONE 64-bit word per rating = (cluster : raw item : slot of the rater : rating as a half); sorted by its top bits alone
__global__ void k_cluster_item_keys_packed(int64_t n, const uint64_t* __restrict__ ukeys, const uint32_t* __restrict__ du1,
                                           const int32_t* __restrict__ ucluster, const int32_t* __restrict__ du2slot, uint64_t* __restrict__ keys, int ib, int sb) {
    const uint64_t mask = ((uint64_t)1 << ib) - 1;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t du = du1[t] - 1;
        const uint64_t uk = ukeys[t];
        keys[t] = ((((uint64_t)(uint32_t)ucluster[du] << ib) | ((uk >> 16) & mask)) << (sb + 16)) | ((uint64_t)(uint32_t)du2slot[du] << 16) | (uk & 0xFFFFu);
    }
}
__global__ void k_heads_packed_dup(int64_t n, const uint64_t* __restrict__ keys, uint32_t* __restrict__ head, int* __restrict__ err, int sb) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const bool same = t > 0 && (keys[t] >> (sb + 16)) == (keys[t - 1] >> (sb + 16));
        head[t] = !same;
        if (same && (keys[t] >> 16) == (keys[t - 1] >> 16)) atomicOr(err, ERR_DUP);      // same column, same slot
    }
}
__global__ void k_fill_csc_packed(int64_t n, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pr1, int sb,
                                  int32_t* __restrict__ csc_slot, float* __restrict__ csc_r, int32_t* __restrict__ csc_pair) {
    const uint64_t smask = ((uint64_t)1 << sb) - 1;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[q];
        csc_slot[q] = (int32_t)((k >> 16) & smask);
        csc_r[q] = fy_key_rating(k);
        csc_pair[q] = (int32_t)(pr1[q] - 1);
    }
}

__global__ void k_scatter_pairs(int64_t n, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ head,
                                const uint32_t* __restrict__ pr1, int32_t* __restrict__ pair_cluster,
                                int32_t* __restrict__ pair_item, int32_t* __restrict__ pair_start, int ib, int low /* packed mode: payload bits below (cluster : item) */) {
    const uint64_t mask = ((uint64_t)1 << ib) - 1;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        if (head[t]) {
            const uint32_t p = pr1[t] - 1;
            const uint64_t k = keys[t] >> low;
            const int32_t c = (int32_t)(k >> ib);
            pair_cluster[p] = c;
            pair_item[p] = (int32_t)(uint32_t)(k & mask);
            pair_start[p] = (int32_t)t;
        }
}

// pairs are cluster-major: pairs of cluster c = [first pair with cluster >= c, first pair with cluster >= c + 1).  (Counting
// them with one atomic per pair -- 59 k atomics on ONE address for a single cluster -- took 0.6 ms.)
__global__ void k_cluster_pair_counts(int32_t K, int32_t nP, const int32_t* __restrict__ pair_cluster, int32_t* __restrict__ pcount) {
    for (int32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < K; c += gridDim.x * blockDim.x) {
        int32_t b[2];
        for (int x = 0; x < 2; x++) {
            int32_t lo = 0, hi = nP;
            while (lo < hi) {
                const int32_t mid = (lo + hi) >> 1;
                if (pair_cluster[mid] < c + x) lo = mid + 1; else hi = mid;
            }
            b[x] = lo;
        }
        pcount[c] = b[1] - b[0];
    }
}

__global__ void k_item_keys_of_pairs(int32_t nP, const int32_t* __restrict__ pair_item, uint64_t* __restrict__ keys,
                                     uint32_t* __restrict__ vals) {
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < nP; p += gridDim.x * blockDim.x) {
        keys[p] = (uint64_t)(uint32_t)pair_item[p];
        vals[p] = (uint32_t)p;
    }
}

__global__ void k_scatter_items(int32_t nP, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                const uint32_t* __restrict__ head, const uint32_t* __restrict__ di1,
                                int32_t* __restrict__ iid, int32_t* __restrict__ pair_di) {
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nP; q += gridDim.x * blockDim.x) {
        const uint32_t d = di1[q] - 1;
        if (head[q]) iid[d] = (int32_t)(uint32_t)keys[q];
        pair_di[vals[q]] = (int32_t)d;
    }
}

// CSC payload in pair order: slot of the rater, its raw rating, and the pair every entry belongs to
__global__ void k_fill_csc(int64_t n, const uint64_t* __restrict__ sorted_vals, const uint32_t* __restrict__ pr1,
                           int32_t* __restrict__ csc_slot, float* __restrict__ csc_r, int32_t* __restrict__ csc_pair) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t v = sorted_vals[q];
        csc_slot[q] = (int32_t)(v >> 32);
        csc_r[q] = __uint_as_float((uint32_t)v);
        csc_pair[q] = (int32_t)(pr1[q] - 1);
    }
}

__global__ void k_rank_keys(int32_t nP, const int32_t* __restrict__ pair_cluster, const int32_t* __restrict__ pair_start,
                            uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < nP; p += gridDim.x * blockDim.x) {
        const uint32_t cnt = (uint32_t)(pair_start[p + 1] - pair_start[p]);
        keys[p] = ((uint64_t)(uint32_t)pair_cluster[p] << 32) | (0xFFFFFFFFu - cnt);
        vals[p] = (uint32_t)p;
    }
}

__global__ void k_scatter_ranks(int32_t nP, const uint32_t* __restrict__ rank_pair_u, const int32_t* __restrict__ pair_cluster,
                                const int32_t* __restrict__ pcstart, const int32_t* __restrict__ pair_di,
                                const int32_t* __restrict__ iid, int32_t* __restrict__ rank_pair,
                                int32_t* __restrict__ pair_rank, int32_t* __restrict__ rank_item_raw) {
    for (int32_t pos = blockIdx.x * blockDim.x + threadIdx.x; pos < nP; pos += gridDim.x * blockDim.x) {
        const int32_t p = (int32_t)rank_pair_u[pos];
        rank_pair[pos] = p;
        pair_rank[p] = pos - pcstart[pair_cluster[p]];
        rank_item_raw[pos] = iid[pair_di[p]];
    }
}

__global__ void k_csr_keys(int64_t n, const int32_t* __restrict__ csc_slot, const int32_t* __restrict__ csc_pair,
                           const int32_t* __restrict__ pair_rank, uint64_t* __restrict__ keys, int rb) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x)
        keys[q] = ((uint64_t)(uint32_t)csc_slot[q] << rb) | (uint32_t)pair_rank[csc_pair[q]];
}

// CSC entries of one (cluster, item) column in RANK order: number of entries of the column at every rank position
__global__ void k_rank_counts(int32_t nP, const int32_t* __restrict__ rank_pair, const int32_t* __restrict__ pair_start, int32_t* __restrict__ cnt) {
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= nP; r += gridDim.x * blockDim.x) {
        const int32_t p = r < nP ? rank_pair[r] : 0;
        cnt[r] = r < nP ? pair_start[p + 1] - pair_start[p] : 0;
    }
}
// the CSC entries moved column by column into rank order (cluster-major, inside a cluster by compact item index), each with its
// key (slot << rb) | index: a STABLE sort by the slot bits alone then leaves every row's indices ascending
__global__ void k_csr_keys_ranked(int64_t n, const int32_t* __restrict__ csc_slot, const int32_t* __restrict__ csc_pair, const float* __restrict__ csc_r,
                                  const int32_t* __restrict__ pair_rank, const int32_t* __restrict__ pair_cluster, const int32_t* __restrict__ pcstart,
                                  const int32_t* __restrict__ pair_start, const int32_t* __restrict__ rank_start, uint64_t* __restrict__ keys,
                                  float* __restrict__ vals, int rb) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int32_t p = csc_pair[q];
        const int32_t idx = pair_rank[p];
        const int64_t dst = (int64_t)rank_start[pcstart[pair_cluster[p]] + idx] + (q - pair_start[p]);
        keys[dst] = ((uint64_t)(uint32_t)csc_slot[q] << rb) | (uint32_t)idx;
        vals[dst] = csc_r[q];
    }
}

__global__ void k_csr_keys_ranked_packed(int64_t n, const int32_t* __restrict__ csc_slot, const int32_t* __restrict__ csc_pair, const float* __restrict__ csc_r,
                                         const int32_t* __restrict__ pair_rank, const int32_t* __restrict__ pair_cluster, const int32_t* __restrict__ pcstart,
                                         const int32_t* __restrict__ pair_start, const int32_t* __restrict__ rank_start, uint64_t* __restrict__ keys, int rb) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int32_t p = csc_pair[q];
        const int32_t idx = pair_rank[p];
        const int64_t dst = (int64_t)rank_start[pcstart[pair_cluster[p]] + idx] + (q - pair_start[p]);
        keys[dst] = ((((uint64_t)(uint32_t)csc_slot[q] << rb) | (uint32_t)idx) << 16) | (uint64_t)__half_as_ushort(__float2half(csc_r[q]));
    }
}
__global__ void k_csr_unpack(int64_t n, const uint64_t* __restrict__ keys, int32_t* __restrict__ idx, float* __restrict__ r, int rb) {
    const uint64_t mask = ((uint64_t)1 << rb) - 1;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[q];
        idx[q] = (int32_t)(uint32_t)((k >> 16) & mask);
        r[q] = fy_key_rating(k);
    }
}
__global__ void k_low_bits(int64_t n, const uint64_t* __restrict__ keys, int32_t* __restrict__ out, int rb) {
    const uint64_t mask = ((uint64_t)1 << rb) - 1;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x)
        out[q] = (int32_t)(uint32_t)(keys[q] & mask);
}

__global__ void k_slot_degrees(int32_t nU, const int32_t* __restrict__ slot2du, const int32_t* __restrict__ udeg,
                               const int32_t* __restrict__ ucluster, const int32_t* __restrict__ pcstart,
                               int32_t* __restrict__ deg_slot, int64_t* __restrict__ work, int64_t* __restrict__ deg2) {
    for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nU; s += gridDim.x * blockDim.x) {
        const int32_t u = slot2du[s];
        const int64_t n = udeg[u];
        const int32_t c = ucluster[u];
        const int64_t Ic = pcstart[c + 1] - pcstart[c];
        deg_slot[s] = (int32_t)n;
        work[s] = n * (Ic - n) + 1;   // +1 keeps the prefix strictly increasing
        deg2[s] = n * n;
    }
}

__global__ void k_gather_i32(int32_t n, const int32_t* __restrict__ src, const int32_t* __restrict__ index, int32_t* __restrict__ dst) {
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) dst[t] = src[index[t]];
}

__global__ void k_set_i32(int32_t* __restrict__ at, int32_t value) { *at = value; }
// pcstart = exclusive prefix of the per-cluster pair counts (K + 1 entries), cluster_q[c] = pair_start[pcstart[c]]: one thread, K is small
__global__ void k_cluster_starts(int32_t K, const int32_t* __restrict__ pcount, const int32_t* __restrict__ pair_start, int32_t* __restrict__ pcstart,
                                 int32_t* __restrict__ cluster_q) {
    if (blockIdx.x || threadIdx.x) return;
    int32_t at = 0;
    for (int32_t c = 0; c <= K; c++) {
        pcstart[c] = at;
        cluster_q[c] = pair_start[at];
        if (c < K) at += pcount[c];
    }
}
__global__ void k_gather_i64(int32_t n, const int64_t* __restrict__ src, const int32_t* __restrict__ index, int64_t* __restrict__ dst) {
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) dst[t] = src[index[t]];
}
// out[t] = src[index[t]] for a host list of positions: ONE gather launch and one copy back (a d2h per element is ~15 us each -- 100
// of them per job at 50 clusters).  Synchronises the context.
void gather_to_host_i32(Context* ctx, const int32_t* src, const std::vector<int32_t>& index, int32_t* out) {
    if (index.empty()) return;
    DevBuf<int32_t> di(ctx, index.size()), dv(ctx, index.size());
    h2d(ctx, di.get(), index.data(), index.size());
    k_gather_i32<<<grid_for((int64_t)index.size()), 256, 0, ctx->stream>>>((int32_t)index.size(), src, di.get(), dv.get());
    FY_KERNEL_CHECK();
    d2h(ctx, out, dv.get(), index.size());
    sync(ctx);
}
void gather_to_host_i64(Context* ctx, const int64_t* src, const std::vector<int32_t>& index, int64_t* out) {
    if (index.empty()) return;
    DevBuf<int32_t> di(ctx, index.size());
    DevBuf<int64_t> dv(ctx, index.size());
    h2d(ctx, di.get(), index.data(), index.size());
    k_gather_i64<<<grid_for((int64_t)index.size()), 256, 0, ctx->stream>>>((int32_t)index.size(), src, di.get(), dv.get());
    FY_KERNEL_CHECK();
    d2h(ctx, out, dv.get(), index.size());
    sync(ctx);
}

// ---------------------------------------------------------------- build
void ratings_id_bounds(Context* ctx, fy_ratings* R) {
    R->max_user = R->max_item = -1;
    R->scores_fp16_exact = false;
    if (R->nnz == 0) return;
    DevBuf<int32_t> max_ids(ctx, 3);
    FY_HIP(hipMemsetAsync(max_ids.get(), 0xFF, 2 * sizeof(int32_t), ctx->stream));   // -1
    FY_HIP(hipMemsetAsync(max_ids.get() + 2, 0, sizeof(int32_t), ctx->stream));
    k_max_ids<<<grid_for(R->nnz), 256, 0, ctx->stream>>>(R->nnz, R->user.get(), R->item.get(), R->score.get(), 2, max_ids.get());
    FY_KERNEL_CHECK();
    int32_t h[3];
    d2h(ctx, h, max_ids.get(), 3);
    sync(ctx);
    R->max_user = h[0];
    R->max_item = h[1];
    R->scores_fp16_exact = h[2] == 0;
}

// ---------------------------------------------------------------- sharded prep: the owned clusters' ratings alone
// (the reference partitions the ratings by cluster on the map side -- M/util/IntKeyPartitioner.java:15, RM2Job.java:130-133, 251 -- so a
// reduce group only ever sorts its own cluster's ratings; a rank that sorted ALL ratings to score an eighth of the clusters did eight
// times the reference's share of that work)

// degree of every raw user over the kept ratings, which raw items are rated at all.  A wave adds a RUN of neighbours with one user
// with one atomic (input grouped by user -- what a file or a Cassandra partition scan delivers -- would otherwise send 64 atomics of a wave
// to one address: 1.1 ms at ML-25M shape; in any other order the atomics go to different addresses and do not queue).
__global__ void k_shard_degrees(int64_t n, const int32_t* __restrict__ user, const int32_t* __restrict__ item, const float* __restrict__ score,
                                int32_t* __restrict__ deg, int32_t* __restrict__ seen, int* __restrict__ err) {
    const int lane = threadIdx.x & 63;
    const int64_t n_up = (n + 63) & ~(int64_t)63;       // whole waves enter every round (ballot / shuffle below)
    bool neg = false;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_up; t += (int64_t)gridDim.x * blockDim.x) {
        int32_t u = -1;
        if (t < n && score[t] > 0.0f) {
            u = user[t];
            const int32_t i = item[t];
            if (u < 0 || i < 0) { neg = true; u = -1; }
            else if (!seen[i]) seen[i] = 1;
        }
        const int32_t before = __shfl_up(u, 1, 64);
        const bool head = u >= 0 && (lane == 0 || before != u);
        const bool breaks = lane == 0 || before != u;                 // a run ends in front of every lane whose user differs
        const unsigned long long bm = __ballot(breaks);
        if (head) {
            const unsigned long long later = lane == 63 ? 0ull : (bm >> (lane + 1));
            const int len = later ? __ffsll((long long)later) : 64 - lane;
            atomicAdd(&deg[u], len);
        }
    }
    if (__ballot(neg) && lane == 0) atomicOr(err, ERR_NEG_ID);
}

// per cluster: ratings, sum of n_u^2, rated users -- counted in LDS per workgroup, integer atomics (the same numbers on every rank);
// own[u] is filled later (k_shard_owned)
constexpr int SHARD_LDS_CLUSTERS = 2048;
__global__ void k_shard_cluster_weights(int32_t n_raw_users, const int32_t* __restrict__ deg, const int32_t* __restrict__ cl_of_raw, int32_t K,
                                        unsigned long long* __restrict__ w /* 3 K + 1 */, const int32_t* __restrict__ seen, int32_t n_raw_items,
                                        int* __restrict__ err) {
    __shared__ unsigned long long sh[3 * SHARD_LDS_CLUSTERS];
    const bool lds = K <= SHARD_LDS_CLUSTERS;
    if (lds) {
        for (int c = threadIdx.x; c < 3 * K; c += blockDim.x) sh[c] = 0;
        __syncthreads();
    }
    const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t u = t0; u < n_raw_users; u += step) {
        const unsigned long long d = (unsigned long long)deg[u];
        if (!d) continue;
        int32_t c = cl_of_raw[u];
        if (c < 0 || c >= K) { atomicOr(err, ERR_CLUSTER_RANGE); continue; }
        unsigned long long* dst = lds ? sh : w;
        atomicAdd(&dst[c], d);
        atomicAdd(&dst[K + c], d * d);
        atomicAdd(&dst[2 * K + c], 1ull);
    }
    unsigned long long items = 0;
    for (int64_t i = t0; i < n_raw_items; i += step) items += seen[i] ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) items += __shfl_down(items, o, 64);
    if ((threadIdx.x & 63) == 0 && items) atomicAdd(&w[3 * K], items);
    if (lds) {
        __syncthreads();
        for (int c = threadIdx.x; c < 3 * K; c += blockDim.x)
            if (sh[c]) atomicAdd(&w[c], sh[c]);
    }
}

// own[u] = 1: raw user u belongs to one of this rank's clusters
__global__ void k_shard_owned(int32_t n_raw_users, const int32_t* __restrict__ cl_of_raw, const int32_t* __restrict__ owner, int32_t rank,
                              uint8_t* __restrict__ own) {
    for (int32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n_raw_users; u += gridDim.x * blockDim.x) own[u] = owner[cl_of_raw[u]] == rank ? 1 : 0;
}

// The owned ratings, compacted in input order without a flag per rating: a wave's 64 keep-bits as one mask and one count (pass 1), an
// exclusive scan of the 390 000 counts, and the kept entries written behind their wave's base (pass 2) -- 5 MB of bookkeeping
// instead of 300 MB of flags and positions.
__global__ void k_shard_masks(int64_t n, const int32_t* __restrict__ user, const float* __restrict__ score, const uint8_t* __restrict__ own,
                              unsigned long long* __restrict__ mask, int32_t* __restrict__ count) {
    const int64_t n_up = (n + 63) & ~(int64_t)63;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_up; t += (int64_t)gridDim.x * blockDim.x) {
        const bool keep = t < n && score[t] > 0.0f && own[user[t]];      // (ids were range-checked by k_shard_degrees)
        const unsigned long long m = __ballot(keep);
        if ((threadIdx.x & 63) == 0) { mask[t >> 6] = m; count[t >> 6] = (int32_t)__popcll(m); }
    }
}
__global__ void k_shard_compact(int64_t n, const unsigned long long* __restrict__ mask, const int32_t* __restrict__ base, const int32_t* __restrict__ user,
                                const int32_t* __restrict__ item, const float* __restrict__ score, int32_t* __restrict__ ou, int32_t* __restrict__ oi,
                                float* __restrict__ os) {
    const int lane = threadIdx.x & 63;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long m = mask[t >> 6];
        if ((m >> lane) & 1ull) {
            const int32_t p = base[t >> 6] + (int32_t)__popcll(m & ((1ull << lane) - 1ull));
            ou[p] = user[t];
            oi[p] = item[t];
            os[p] = score[t];
        }
    }
}

// cuts `w` (one weight per position, in order) into `parts` contiguous runs with the smallest possible largest run (binary search on the
// cap, greedy fill); returns the first position of every run (parts + 1 entries; trailing runs may be empty)
static std::vector<int> linear_partition(const std::vector<int64_t>& w, int parts) {
    const int n = (int)w.size();
    auto runs_needed = [&](int64_t cap, std::vector<int>* first) -> int {
        int runs = 1;
        int64_t in_run = 0;
        if (first) { first->clear(); first->push_back(0); }
        for (int k = 0; k < n; k++) {
            if (w[(size_t)k] > cap) return parts + 1;
            if (in_run + w[(size_t)k] > cap && in_run > 0) { runs++; in_run = 0; if (first) first->push_back(k); }
            in_run += w[(size_t)k];
        }
        return runs;
    };
    int64_t lo = 0, hi = 0;
    for (int64_t x : w) hi += x;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (runs_needed(mid, nullptr) <= parts) hi = mid; else lo = mid + 1;
    }
    std::vector<int> first;
    runs_needed(hi, &first);
    while ((int)first.size() < parts + 1) first.push_back(n);
    return first;
}

bool shard_ratings_by_cluster(Context* ctx, const fy_ratings* R, int32_t K, int64_t n_map, const int32_t* map_user, const int32_t* map_cluster,
                              int rank, int world, fy_ratings& mine, std::vector<int32_t>& owner, DevBuf<int32_t>& d_cl) {
    if (world <= 1 || K < world || R->nnz == 0 || R->max_user < 0 || R->max_item < 0) return false;
    if (R->nnz >= (int64_t)0x7FFFFFF0) return false;          // (build_structure reports it)
    const int64_t nRU = (int64_t)R->max_user + 1, nRI = (int64_t)R->max_item + 1;
    if (nRU + nRI > ((int64_t)1 << 25)) return false;           // the exchange buffer is indexed by raw ids in this mode: 256 MB of doubles at most
    // a cluster nobody of the map is routed to can only be cluster 0 (the users the map does not name): a quick host-side bound on the
    // number of non-empty clusters before any device work
    std::vector<int32_t> cl_of_raw((size_t)nRU, 0);
    {
        std::vector<char> named((size_t)K, 0);
        int n_named = 1;        // cluster 0
        named[0] = 1;
        for (int64_t m = 0; m < n_map; m++) {
            const int32_t u = map_user[m], c = map_cluster[m];
            if (u < 0 || u >= nRU) continue;
            cl_of_raw[(size_t)u] = c;                             // a later pair of the same user wins (successive TIntIntHashMap.put calls)
            if (c >= 0 && c < K && !named[(size_t)c]) { named[(size_t)c] = 1; n_named++; }
        }
        if (n_named < world) return false;
    }
    hipStream_t st = ctx->stream;
    SyncOnUnwind guard(st);
    DevBuf<int32_t> deg(ctx, (size_t)nRU), seen(ctx, (size_t)nRI);
    d_cl.alloc(ctx, (size_t)nRU);
    DevBuf<unsigned long long> w(ctx, 3 * (size_t)K + 1);
    DevBuf<int> err(ctx, 1);
    h2d(ctx, d_cl.get(), cl_of_raw.data(), (size_t)nRU);
    deg.zero();
    seen.zero();
    w.zero();
    err.zero();
    k_shard_degrees<<<grid_for(R->nnz), 256, 0, st>>>(R->nnz, R->user.get(), R->item.get(), R->score.get(), deg.get(), seen.get(), err.get());
    FY_KERNEL_CHECK();
    k_shard_cluster_weights<<<grid_for(std::max(nRU, nRI)), 256, 0, st>>>((int32_t)nRU, deg.get(), d_cl.get(), K, w.get(), seen.get(), (int32_t)nRI, err.get());
    FY_KERNEL_CHECK();
    std::vector<unsigned long long> hw(3 * (size_t)K + 1);
    int h_err = 0;
    d2h(ctx, hw.data(), w.get(), hw.size());
    d2h(ctx, &h_err, err.get(), 1);
    sync(ctx);
    // (the failures every rank sees the same way are reported here, by every rank; the rank-local ones -- a duplicate rating, a
    // clusteringCount mismatch inside an owned cluster -- travel with the statistics, fy_rm2.hip)
    if (h_err & ERR_NEG_ID) FY_FAIL(FY_ERR_NEGATIVE_ID, "negative user or item id in the ratings");
    if (h_err & ERR_CLUSTER_RANGE) FY_FAIL(FY_ERR_CLUSTER_RANGE, "a rated user is routed to a cluster outside [0, %d)", K);
    std::vector<int32_t> ids;       // the non-empty clusters, ascending
    std::vector<int64_t> weight;
    const int64_t items_all = (int64_t)hw[3 * (size_t)K];
    for (int c = 0; c < K; c++)
        if (hw[2 * (size_t)K + c]) {
            ids.push_back(c);
            // what a whole cluster costs its rank: the co-rating pairs of the row kernel (sum of n_u^2) + the log terms of scoring, at
            // most (ratings of the cluster) x (items anybody rated)
            weight.push_back((int64_t)hw[(size_t)K + c] + (int64_t)hw[(size_t)c] * std::min<int64_t>(items_all, (int64_t)hw[(size_t)c]));
        }
    if ((int)ids.size() < world) return false;
    const std::vector<int> first = linear_partition(weight, world);
    owner.assign((size_t)K, -1);
    for (int r = 0; r < world; r++)
        for (int k = first[(size_t)r]; k < first[(size_t)r + 1]; k++) owner[(size_t)ids[(size_t)k]] = r;
    for (int c = 0; c < K; c++)
        if (owner[(size_t)c] < 0) owner[(size_t)c] = world;     // empty clusters: nobody's
    DevBuf<int32_t> d_owner(ctx, (size_t)K);
    h2d(ctx, d_owner.get(), owner.data(), (size_t)K);
    DevBuf<uint8_t> own(ctx, (size_t)nRU);
    k_shard_owned<<<grid_for(nRU), 256, 0, st>>>((int32_t)nRU, d_cl.get(), d_owner.get(), rank, own.get());
    FY_KERNEL_CHECK();
    const int64_t n_waves = ceil_div(R->nnz, 64);
    DevBuf<unsigned long long> mask(ctx, (size_t)n_waves);
    DevBuf<int32_t> count(ctx, (size_t)n_waves + 1), base(ctx, (size_t)n_waves + 1);
    k_shard_masks<<<grid_for(R->nnz), 256, 0, st>>>(R->nnz, R->user.get(), R->score.get(), own.get(), mask.get(), count.get());
    FY_KERNEL_CHECK();
    k_set_i32<<<1, 1, 0, st>>>(count.get() + n_waves, 0);
    FY_KERNEL_CHECK();
    exclusive_scan_i32(ctx, count.get(), base.get(), (size_t)n_waves + 1);
    const int64_t kept = (int64_t)fetch(ctx, base.get() + n_waves);
    mine.ctx = ctx;
    mine.nnz = kept;
    mine.max_user = R->max_user;
    mine.max_item = R->max_item;
    mine.scores_fp16_exact = R->scores_fp16_exact;
    mine.user.alloc(ctx, (size_t)kept);
    mine.item.alloc(ctx, (size_t)kept);
    mine.score.alloc(ctx, (size_t)kept);
    if (kept) {
        k_shard_compact<<<grid_for(R->nnz), 256, 0, st>>>(R->nnz, mask.get(), base.get(), R->user.get(), R->item.get(), R->score.get(), mine.user.get(),
                                                          mine.item.get(), mine.score.get());
        FY_KERNEL_CHECK();
    }
    sync(ctx);      // (the masks and the tables are released here)
    return true;
}

void build_structure(Context* ctx, const fy_ratings* R, int32_t K, int64_t n_map, const int32_t* map_user,
                     const int32_t* map_cluster, const int32_t* cluster_count, bool keep_nonpositive, Prepared& P, const int32_t* cl_of_raw) {
    P.ctx = ctx;
    P.K = K;
    const int64_t n_in = R->nnz;
    if (n_in >= (int64_t)0x7FFFFFF0) FY_FAIL(FY_ERR_UNSUPPORTED, "nnz = %lld exceeds the 2^31 offset limit", (long long)n_in);
    hipStream_t st = ctx->stream;

    DevBuf<int> err(ctx, 1);
    err.zero();
    DevBuf<unsigned long long> kept(ctx, 1);
    kept.zero();

    // ---- sort #1: user-major order, duplicates / negative ids detected
    DevBuf<uint64_t> k1a(ctx, n_in), k1b(ctx, n_in);
    DevBuf<float> sc_um;
    auto bits_for = [](uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; };   // bits that hold every value <= v
    int ib = 1;   // bits of the item field of the sort keys
    // PACKED mode (round 4): every score is exactly a half (found when the ratings entered HBM) and (user : item), (cluster : item : slot)
    // and (slot : index) each leave 16 bits of a 64-bit word free: the rating rides in the low 16 bits of every sort key and the three
    // nnz-sized sorts move KEYS ALONE -- 16 bytes per rating and pass instead of 24 / 32 / 24.
    int pb = 0;
    if (n_in) {
        const int ib0 = std::max(1, bits_for((uint64_t)std::max(0, R->max_item))), ub0 = std::max(1, bits_for((uint64_t)(R->max_user + 1)));
        const int kb0 = bits_for((uint64_t)std::max(0, K - 1));
        // (slots and item indices never need more bits than the user / item ids they number)
        if (ctx->tune.prep_packed && R->scores_fp16_exact && ib0 + ub0 + 16 <= 64 && kb0 + ib0 + ub0 + 16 <= 64) pb = 16;
    }
    if (!pb) sc_um.alloc(ctx, n_in);
    if (n_in) {
        // (the id bounds were found when the ratings entered HBM, fy_ratings_create: a property of the container like nnz)
        const int32_t hmax[2] = {R->max_user, R->max_item};
        const uint32_t drop_user = (uint32_t)(hmax[0] + 1);
        ib = std::max(1, bits_for((uint64_t)std::max(0, hmax[1])));
        const int ub = std::max(1, bits_for(drop_user));
        k_user_item_keys<<<grid_for(n_in), 256, 0, st>>>(n_in, R->user.get(), R->item.get(), R->score.get(),
                                                          keep_nonpositive ? 1 : 0, ib, drop_user, k1a.get(), kept.get(), err.get(), pb);
        FY_KERNEL_CHECK();
        // by the USER bits alone (a stable sort: inside a user the ratings keep the order of the input): 3 radix passes instead of 5 at
        // ML-25M shape.  Nothing needs the items of a user in order here -- the CSR is sorted by (slot, item index) below -- except the
        // duplicate check, which moved behind the (cluster, item) sort, where one user's two ratings of an item are neighbours too.
        if (pb) sort_keys_u64(ctx, k1a.get(), k1b.get(), n_in, std::min(64, ib + pb + ub), ib + pb);
        else sort_pairs_u64_f32(ctx, k1a.get(), k1b.get(), const_cast<float*>(R->score.get()), sc_um.get(), n_in, std::min(64, ib + ub), ib);
    }
    // (host round trips cost ~30 us each plus the bubble behind them: what can be read together is read together)
    unsigned long long h_kept = 0;
    int h_flags = 0;
    d2h(ctx, &h_kept, kept.get(), 1);
    d2h(ctx, &h_flags, err.get(), 1);
    sync(ctx);
    const int64_t nnz = (int64_t)h_kept;
    P.nnz = nnz;
    {
        const int flags = h_flags;
        if (flags & ERR_NEG_ID) FY_FAIL(FY_ERR_NEGATIVE_ID, "negative user or item id in the ratings");
        P.ratings_fp16_exact = !(flags & NOTE_NOT_FP16);
        P.ratings_positive = !(flags & NOTE_NONPOSITIVE);
        P.ratings_frac_bits = 0;
        for (int m = 0; m <= 9; m++)
            if (flags & (1 << (8 + m))) P.ratings_frac_bits = m;
    }
    if (nnz == 0) {
        P.nU = P.nI = P.nP = 0;
        P.csize.assign(K, 0);
        P.ucstart.assign(K + 1, 0);
        P.pcstart.assign(K + 1, 0);
        P.cluster_q.assign(K + 1, 0);
        return;
    }
    k1a.release();
    uint64_t* ukeys = k1b.get();   // user-major (items of a user in input order), first nnz entries are the kept ratings

    DevBuf<uint32_t> head(ctx, nnz), du1(ctx, nnz);
    k_heads_hi32<<<grid_for(nnz), 256, 0, st>>>(nnz, ukeys, head.get(), 0, err.get(), ib + pb);
    FY_KERNEL_CHECK();
    inclusive_scan_u32(ctx, head.get(), du1.get(), nnz);
    const int32_t nU = (int32_t)fetch(ctx, du1.get() + (nnz - 1));
    P.nU = nU;

    P.uid.alloc(ctx, nU);
    DevBuf<int32_t> ustart(ctx, (size_t)nU + 1);
    k_scatter_users<<<grid_for(nnz), 256, 0, st>>>(nnz, ukeys, head.get(), du1.get(), P.uid.get(), ustart.get(), ib + pb);
    FY_KERNEL_CHECK();
    k_set_i32<<<1, 1, 0, st>>>(ustart.get() + nU, (int32_t)nnz);
    FY_KERNEL_CHECK();
    P.usum.alloc(ctx, nU);
    P.udeg.alloc(ctx, nU);
    k_user_sums<<<grid_for((int64_t)nU * 64, 256), 256, 0, st>>>(nU, ustart.get(), sc_um.get(), pb ? ukeys : nullptr, P.usum.get(), P.udeg.get());
    FY_KERNEL_CHECK();

    // ---- cluster routing
    P.ucluster.alloc(ctx, nU);
    std::vector<uint64_t> hm;      // (host sources of two uploads: they live until the next synchronisation, below)
    std::vector<uint32_t> hv;
    std::vector<int32_t> h_table;
    SyncOnUnwind hm_hv_guard(st);      // an allocation or a sort below may throw while the uploads are still queued
    // Dense user ids (the table is at most a few times the map): cluster by raw id from a table -- filled on the host in file order (a later
    // pair of one user wins, like successive TIntIntHashMap.put calls; users the map does not name stay in cluster 0), one upload, no
    // sort.  The sharded prep hands its table over.
    DevBuf<int32_t> d_table_own;
    const int64_t nRU = (int64_t)R->max_user + 1;
    if (!cl_of_raw && n_map > 0 && nRU <= 8 * n_map + (1 << 20)) {
        h_table.assign((size_t)nRU, 0);
        for (int64_t m = 0; m < n_map; m++)
            if (map_user[m] >= 0 && map_user[m] < nRU) h_table[(size_t)map_user[m]] = map_cluster[m];
        d_table_own.alloc(ctx, (size_t)nRU);
        h2d(ctx, d_table_own.get(), h_table.data(), (size_t)nRU);
        cl_of_raw = d_table_own.get();
    }
    if (cl_of_raw) {
        k_lookup_cluster_table<<<grid_for(nU), 256, 0, st>>>(nU, P.uid.get(), cl_of_raw, K, P.ucluster.get(), err.get());
        FY_KERNEL_CHECK();
    } else {
        if (n_map >= ((int64_t)1 << 32)) FY_FAIL(FY_ERR_UNSUPPORTED, "clustering map with more than 2^32 entries");
        hm.resize((size_t)n_map);
        hv.resize((size_t)n_map);
        for (int64_t m = 0; m < n_map; m++) {
            // key = (user : position in the file): sorted by it, the pairs of one user keep their order -- a later pair of the same
            // user wins, like successive TIntIntHashMap.put calls.  (A negative id can never match a kept rating.)
            hm[m] = ((uint64_t)(map_user[m] < 0 ? 0xFFFFFFFFu : (uint32_t)map_user[m]) << 32) | (uint32_t)m;
            hv[m] = (uint32_t)map_cluster[m];
        }
        // sorted on the device: std::stable_sort of 162 541 pairs on the host was 3-4 ms of every cold multi-cluster prepare
        DevBuf<uint64_t> dm_in(ctx, (size_t)n_map), dm(ctx, (size_t)n_map);
        DevBuf<uint32_t> dv_in(ctx, (size_t)n_map), dv(ctx, (size_t)n_map);
        h2d(ctx, dm_in.get(), hm.data(), (size_t)n_map);
        h2d(ctx, dv_in.get(), hv.data(), (size_t)n_map);
        sort_pairs_u64_u32(ctx, dm_in.get(), dm.get(), dv_in.get(), dv.get(), (size_t)n_map);
        k_lookup_cluster<<<grid_for(nU), 256, 0, st>>>(nU, P.uid.get(), n_map, dm.get(), dv.get(), K, P.ucluster.get(), err.get());
        FY_KERNEL_CHECK();
    }
    // (a user routed outside [0, K) is counted in cluster 0 by the kernels below and reported at the next synchronisation)

    // ---- slots: cluster-major, heavy rows first
    P.d_csize.alloc(ctx, (size_t)K);
    P.d_csize.zero();
    {
        DevBuf<uint64_t> ka(ctx, nU), kb(ctx, nU);
        DevBuf<uint32_t> va(ctx, nU), vb(ctx, nU);
        if (K > 1 && K <= SLOT_KEYS_LDS)
            k_slot_keys_lds<<<grid_for(nU, 256, 64), 256, 0, st>>>(nU, K, P.ucluster.get(), P.udeg.get(), ka.get(), va.get(), P.d_csize.get());
        else
            k_slot_keys<<<grid_for(nU), 256, 0, st>>>(nU, P.ucluster.get(), P.udeg.get(), ka.get(), va.get(), P.d_csize.get());
        FY_KERNEL_CHECK();
        sort_pairs_u64_u32(ctx, ka.get(), kb.get(), va.get(), vb.get(), nU);
        P.slot2du.alloc(ctx, nU);
        P.du2slot.alloc(ctx, nU);
        k_invert_perm<<<grid_for(nU), 256, 0, st>>>(nU, vb.get(), P.slot2du.get(), P.du2slot.get());
        FY_KERNEL_CHECK();
    }
    P.csize.resize(K);
    d2h(ctx, P.csize.data(), P.d_csize.get(), (size_t)K);
    d2h(ctx, &h_flags, err.get(), 1);
    sync(ctx);
    if (h_flags & ERR_CLUSTER_RANGE) FY_FAIL(FY_ERR_CLUSTER_RANGE, "a rated user is routed to a cluster outside [0, %d)", K);
    P.ucstart.assign(K + 1, 0);
    for (int c = 0; c < K; c++) P.ucstart[c + 1] = P.ucstart[c] + P.csize[c];
    if (cluster_count)
        for (int c = 0; c < K; c++)
            if (P.csize[c] && cluster_count[c] != P.csize[c])
                FY_FAIL(FY_ERR_CLUSTER_COUNT, "clusteringCount[%d] = %d but %d rated users are routed to that cluster", c,
                        cluster_count[c], P.csize[c]);
    P.d_ucstart.alloc(ctx, (size_t)K + 1);
    h2d(ctx, P.d_ucstart.get(), P.ucstart.data(), (size_t)K + 1);

    // ---- sort #3: (cluster, item) order = the CSC
    DevBuf<uint64_t> k3b(ctx, nnz);
    DevBuf<uint64_t> v_sorted;
    const int sb = std::max(1, bits_for((uint64_t)(nU - 1)));      // bits of a slot
    if (pb) {
        DevBuf<uint64_t> k3a(ctx, nnz);
        k_cluster_item_keys_packed<<<grid_for(nnz), 256, 0, st>>>(nnz, ukeys, du1.get(), P.ucluster.get(), P.du2slot.get(), k3a.get(), ib, sb);
        FY_KERNEL_CHECK();
        sort_keys_u64(ctx, k3a.get(), k3b.get(), nnz, std::min(64, sb + 16 + ib + bits_for((uint64_t)(K - 1))), sb + 16);
    } else {
        v_sorted.alloc(ctx, nnz);
        DevBuf<uint64_t> k3a(ctx, nnz), v3a(ctx, nnz);
        k_cluster_item_keys<<<grid_for(nnz), 256, 0, st>>>(nnz, ukeys, du1.get(), P.ucluster.get(), P.du2slot.get(), sc_um.get(), k3a.get(),
                                                            v3a.get(), ib);
        FY_KERNEL_CHECK();
        sort_pairs_u64_u64(ctx, k3a.get(), k3b.get(), v3a.get(), v_sorted.get(), nnz, std::min(64, ib + bits_for((uint64_t)(K - 1))));
    }
    DevBuf<uint32_t> pr1(ctx, nnz);
    if (pb) k_heads_packed_dup<<<grid_for(nnz), 256, 0, st>>>(nnz, k3b.get(), head.get(), err.get(), sb);
    else k_heads_full_dup<<<grid_for(nnz), 256, 0, st>>>(nnz, k3b.get(), v_sorted.get(), head.get(), err.get());
    FY_KERNEL_CHECK();
    inclusive_scan_u32(ctx, head.get(), pr1.get(), nnz);
    uint32_t h_nP = 0;
    d2h(ctx, &h_nP, pr1.get() + (nnz - 1), 1);
    d2h(ctx, &h_flags, err.get(), 1);
    sync(ctx);
    if (h_flags & ERR_DUP) FY_FAIL(FY_ERR_DUPLICATE_RATING, "two ratings share one (user, item) key");
    const int32_t nP = (int32_t)h_nP;
    P.nP = nP;
    P.pair_cluster.alloc(ctx, nP);
    P.pair_start.alloc(ctx, (size_t)nP + 1);
    DevBuf<int32_t> pair_item(ctx, nP), pcount(ctx, (size_t)K);
    k_scatter_pairs<<<grid_for(nnz), 256, 0, st>>>(nnz, k3b.get(), head.get(), pr1.get(), P.pair_cluster.get(),
                                                    pair_item.get(), P.pair_start.get(), ib, pb ? sb + 16 : 0);
    FY_KERNEL_CHECK();
    k_cluster_pair_counts<<<grid_for(K), 256, 0, st>>>(K, nP, P.pair_cluster.get(), pcount.get());
    FY_KERNEL_CHECK();
    k_set_i32<<<1, 1, 0, st>>>(P.pair_start.get() + nP, (int32_t)nnz);
    FY_KERNEL_CHECK();
    {   // first pair of every cluster (prefix of the counts) and its first CSC entry = pair_start there (pairs are cluster-major): on the
        // device, one copy back for both
        P.d_pcstart.alloc(ctx, (size_t)K + 1);
        DevBuf<int32_t> dq(ctx, (size_t)K + 1);
        k_cluster_starts<<<1, 64, 0, st>>>(K, pcount.get(), P.pair_start.get(), P.d_pcstart.get(), dq.get());
        FY_KERNEL_CHECK();
        P.pcstart.assign((size_t)K + 1, 0);
        P.cluster_q.resize((size_t)K + 1);
        d2h(ctx, P.pcstart.data(), P.d_pcstart.get(), (size_t)K + 1);
        d2h(ctx, P.cluster_q.data(), dq.get(), (size_t)K + 1);
        sync(ctx);
    }

    // ---- dense item index (ascending raw id)
    P.pair_di.alloc(ctx, nP);
    {
        DevBuf<uint64_t> ka(ctx, nP), kb(ctx, nP);
        DevBuf<uint32_t> va(ctx, nP), vb(ctx, nP), hd(ctx, nP), di1(ctx, nP);
        k_item_keys_of_pairs<<<grid_for(nP), 256, 0, st>>>(nP, pair_item.get(), ka.get(), va.get());
        FY_KERNEL_CHECK();
        sort_pairs_u64_u32(ctx, ka.get(), kb.get(), va.get(), vb.get(), nP, 32);
        k_heads_full<<<grid_for(nP), 256, 0, st>>>(nP, kb.get(), hd.get());
        FY_KERNEL_CHECK();
        inclusive_scan_u32(ctx, hd.get(), di1.get(), nP);
        P.nI = (int32_t)fetch(ctx, di1.get() + (nP - 1));
        P.iid.alloc(ctx, P.nI);
        k_scatter_items<<<grid_for(nP), 256, 0, st>>>(nP, kb.get(), vb.get(), hd.get(), di1.get(), P.iid.get(), P.pair_di.get());
        FY_KERNEL_CHECK();
    }

    // ---- CSC payload
    P.csc_slot.alloc(ctx, nnz);
    P.csc_r.alloc(ctx, nnz);
    P.csc_pair.alloc(ctx, nnz);
    if (pb) k_fill_csc_packed<<<grid_for(nnz), 256, 0, st>>>(nnz, k3b.get(), pr1.get(), sb, P.csc_slot.get(), P.csc_r.get(), P.csc_pair.get());
    else k_fill_csc<<<grid_for(nnz), 256, 0, st>>>(nnz, v_sorted.get(), pr1.get(), P.csc_slot.get(), P.csc_r.get(), P.csc_pair.get());
    FY_KERNEL_CHECK();

    // ---- popularity rank inside the cluster = compact item index
    P.rank_pair.alloc(ctx, nP);
    P.pair_rank.alloc(ctx, nP);
    P.rank_item_raw.alloc(ctx, nP);
    {
        DevBuf<uint64_t> ka(ctx, nP), kb(ctx, nP);
        DevBuf<uint32_t> va(ctx, nP), vb(ctx, nP);
        k_rank_keys<<<grid_for(nP), 256, 0, st>>>(nP, P.pair_cluster.get(), P.pair_start.get(), ka.get(), va.get());
        FY_KERNEL_CHECK();
        sort_pairs_u64_u32(ctx, ka.get(), kb.get(), va.get(), vb.get(), nP);
        k_scatter_ranks<<<grid_for(nP), 256, 0, st>>>(nP, vb.get(), P.pair_cluster.get(), P.d_pcstart.get(), P.pair_di.get(),
                                                       P.iid.get(), P.rank_pair.get(), P.pair_rank.get(), P.rank_item_raw.get());
        FY_KERNEL_CHECK();
    }

    // ---- sort #6: CSR in slot order, compact index ascending inside a row
    {
        DevBuf<uint64_t> ka(ctx, nnz), kb(ctx, nnz);
        int32_t max_Ic = 1;
        for (int c = 0; c < K; c++) max_Ic = std::max(max_Ic, P.pcstart[c + 1] - P.pcstart[c]);
        const int rb = std::max(1, bits_for((uint64_t)(max_Ic - 1)));   // key = (slot << rb) | compact item index
        // The columns are first moved into rank order (a permutation of whole columns: coalesced), then ONE stable sort by the slot
        // bits alone puts the rows together with their indices ascending: 3 radix passes instead of 5 over (slot, index) at ML-25M shape.
        DevBuf<int32_t> rank_cnt(ctx, (size_t)nP + 1), rank_start(ctx, (size_t)nP + 1);
        k_rank_counts<<<grid_for((int64_t)nP + 1), 256, 0, st>>>(nP, P.rank_pair.get(), P.pair_start.get(), rank_cnt.get());
        FY_KERNEL_CHECK();
        exclusive_scan_i32(ctx, rank_cnt.get(), rank_start.get(), (size_t)nP + 1);
        P.csr_r.alloc(ctx, nnz);
        P.csr_idx.alloc(ctx, nnz);
        if (pb && rb + sb + 16 <= 64) {
            k_csr_keys_ranked_packed<<<grid_for(nnz), 256, 0, st>>>(nnz, P.csc_slot.get(), P.csc_pair.get(), P.csc_r.get(), P.pair_rank.get(), P.pair_cluster.get(),
                                                                     P.d_pcstart.get(), P.pair_start.get(), rank_start.get(), ka.get(), rb);
            FY_KERNEL_CHECK();
            sort_keys_u64(ctx, ka.get(), kb.get(), nnz, std::min(64, 16 + rb + sb), 16 + rb);
            k_csr_unpack<<<grid_for(nnz), 256, 0, st>>>(nnz, kb.get(), P.csr_idx.get(), P.csr_r.get(), rb);
            FY_KERNEL_CHECK();
        } else {
            DevBuf<float> r_ranked(ctx, nnz);
            k_csr_keys_ranked<<<grid_for(nnz), 256, 0, st>>>(nnz, P.csc_slot.get(), P.csc_pair.get(), P.csc_r.get(), P.pair_rank.get(), P.pair_cluster.get(),
                                                              P.d_pcstart.get(), P.pair_start.get(), rank_start.get(), ka.get(), r_ranked.get(), rb);
            FY_KERNEL_CHECK();
            sort_pairs_u64_f32(ctx, ka.get(), kb.get(), r_ranked.get(), P.csr_r.get(), nnz, std::min(64, rb + sb), rb);
            k_low_bits<<<grid_for(nnz), 256, 0, st>>>(nnz, kb.get(), P.csr_idx.get(), rb);
            FY_KERNEL_CHECK();
        }
    }

    // ---- row pointers, work model
    {
        DevBuf<int32_t> deg_slot(ctx, (size_t)nU + 1);
        DevBuf<int64_t> work(ctx, nU), deg2(ctx, nU), wpre(ctx, nU), d2pre(ctx, nU);
        k_slot_degrees<<<grid_for(nU), 256, 0, st>>>(nU, P.slot2du.get(), P.udeg.get(), P.ucluster.get(), P.d_pcstart.get(),
                                                      deg_slot.get(), work.get(), deg2.get());
        FY_KERNEL_CHECK();
        FY_HIP(hipMemsetAsync(deg_slot.get() + nU, 0, sizeof(int32_t), st));
        P.rowptr.alloc(ctx, (size_t)nU + 1);
        exclusive_scan_i32(ctx, deg_slot.get(), P.rowptr.get(), (size_t)nU + 1);
        inclusive_scan_i64(ctx, work.get(), wpre.get(), nU);
        inclusive_scan_i64(ctx, deg2.get(), d2pre.get(), nU);
        P.work_prefix.resize(nU);
        d2h(ctx, P.work_prefix.data(), wpre.get(), (size_t)nU);
        d2h(ctx, &P.sum_deg2, d2pre.get() + (nU - 1), 1);
        std::vector<int64_t> at_end((size_t)K + 1, 0);     // inclusive prefix of n_u^2 at the last slot of every cluster
        {
            std::vector<int32_t> where, which;
            for (int c = 0; c < K; c++)
                if (P.ucstart[c + 1] > P.ucstart[c]) { where.push_back(P.ucstart[c + 1] - 1); which.push_back(c + 1); }
            std::vector<int64_t> got(where.size());
            gather_to_host_i64(ctx, d2pre.get(), where, got.data());      // (synchronises: the copies above have landed too)
            for (size_t t = 0; t < where.size(); t++) at_end[(size_t)which[t]] = got[t];
            if (where.empty()) sync(ctx);
        }
        P.cluster_deg2.assign((size_t)K, 0);
        int64_t before = 0;
        for (int c = 0; c < K; c++) {
            if (P.ucstart[c + 1] > P.ucstart[c]) { P.cluster_deg2[c] = at_end[(size_t)c + 1] - before; before = at_end[(size_t)c + 1]; }
        }
    }
}

void rank_slot_range(const Prepared& P, int rank, int world, int32_t& lo, int32_t& hi) {
    const int32_t n = P.nU;
    if (world <= 1 || n == 0) { lo = 0; hi = n; return; }
    const int64_t total = P.work_prefix[n - 1];
    // At least as many non-empty clusters as ranks: WHOLE clusters per rank (the reference runs one reduce group per
    // cluster, RM2Job.java:251) -- no cluster's co-rating matrix is then built on two ranks and nothing but the item statistics is
    // exchanged.  Ownership stays a contiguous slot range (slots are cluster-major), so the clusters are cut into `world` contiguous
    // runs of the cluster order with the smallest possible largest run (linear partition: binary search on the cap, greedy fill).
    // Every rank computes the same cuts from the same prefix.
    {
        std::vector<int32_t> ends;          // last slot + 1 of every non-empty cluster, ascending
        for (int c = 0; c < P.K; c++)
            if (P.ucstart[c + 1] > P.ucstart[c]) ends.push_back(P.ucstart[c + 1]);
        if ((int)ends.size() >= world) {
            auto upto = [&](int32_t slot_end) -> int64_t { return slot_end > 0 ? P.work_prefix[slot_end - 1] : 0; };
            auto runs_needed = [&](int64_t cap, std::vector<int32_t>* cuts) -> int {
                int runs = 1;
                int64_t start = 0;      // work in front of the current run
                int32_t prev_end = 0;
                if (cuts) cuts->clear();
                for (size_t k = 0; k < ends.size(); k++) {
                    const int64_t here = upto(ends[k]);
                    if (here - start > cap && prev_end > 0 && upto(prev_end) > start) {     // close the run in front of this cluster
                        if (cuts) cuts->push_back(prev_end);
                        runs++;
                        start = upto(prev_end);
                    }
                    if (here - start > cap) return world + 1;                                // a single cluster above the cap
                    prev_end = ends[k];
                }
                return runs;
            };
            int64_t lo_cap = 0, hi_cap = total;
            while (lo_cap < hi_cap) {
                const int64_t mid = lo_cap + (hi_cap - lo_cap) / 2;
                if (runs_needed(mid, nullptr) <= world) hi_cap = mid; else lo_cap = mid + 1;
            }
            std::vector<int32_t> cuts;
            runs_needed(hi_cap, &cuts);
            // fewer runs than ranks (one huge cluster among small ones): the last ranks stay empty rather than split a cluster
            auto cut_at = [&](int r) -> int32_t {
                if (r <= 0) return 0;
                if (r >= world) return n;
                return r - 1 < (int)cuts.size() ? cuts[(size_t)r - 1] : n;
            };
            lo = cut_at(rank);
            hi = cut_at(rank + 1);
            return;
        }
    }
    auto cut = [&](int r) -> int32_t {
        if (r <= 0) return 0;
        if (r >= world) return n;
        // first slot whose inclusive prefix exceeds r/world of the work
        const __int128 target = (__int128)total * r / world;
        return (int32_t)(std::upper_bound(P.work_prefix.begin(), P.work_prefix.end(), (int64_t)target) - P.work_prefix.begin());
    };
    lo = cut(rank);
    hi = cut(rank + 1);
}

}  // namespace fy
