// fy_nmf.hip -- NMF / PPC multiplicative updates that produce the factor matrix H the clustering stage reads
// (second half).
//
// Replaces, per iteration (M = es/udc/fi/dc/irlab/nmf):
//   M/hcomputation/ComputeHJob.java:88-96 (4 MapReduce jobs) + HComputationReducer.java:57-75
//   M/ppc/hcomputation/PPCComputeHJob.java + PPCHComputationReducer.java:61-96
//   M/wcomputation/ComputeWJob.java:88-96 (5 jobs) + WComputationMapper.java:100-118
//   driven by M/AbstractNMFDriver.java:118-146 (H2 and W2 both from the OLD H, W; then swap)
// Everything in fp64 like the reference.  Per iteration: two SpMMs over the ratings (CSR by user against W, CSC by item
// against H: 12 B per rating + one k-row of the dense factor per rating), two k x k Gram matrices (fixed-order two-stage
// reduction: bit-reproducible), two row updates.  HBM-bound on the factor rows gathered by the SpMMs.
#include <algorithm>
#include <cfloat>
#include <vector>

#include "fy_common.hpp"
#include "fy_prep.hpp"
#include "fy_rm2.hpp"

namespace fy {

static inline int grid_for(int64_t n, int block = 256, int cap = 256 * 16) {
    int64_t g = ceil_div(n, block);
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

constexpr int NMF_MAX_K = 256;
constexpr int NMF_GRAM_BLOCKS = 256;

// keys (row << 32 | col) of the kept ratings (score > 0, VectorByItemHDFSMapper.java:37-40); dropped ones sort to the end
__global__ void k_nmf_keys(int64_t n, const int32_t* __restrict__ user, const int32_t* __restrict__ item, const float* __restrict__ score,
                           int32_t n_users, int32_t n_items, uint64_t* __restrict__ by_user, uint64_t* __restrict__ by_item,
                           unsigned long long* __restrict__ kept, int* __restrict__ err) {
    unsigned long long local = 0;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        uint64_t ku = ~0ull, ki = ~0ull;
        if (score[t] > 0.0f) {
            const int32_t u = user[t], i = item[t];
            if (u < 1 || u > n_users || i < 1 || i > n_items) atomicOr(err, 1);
            else {
                ku = ((uint64_t)(uint32_t)(u - 1) << 32) | (uint32_t)(i - 1);
                ki = ((uint64_t)(uint32_t)(i - 1) << 32) | (uint32_t)(u - 1);
                local++;
            }
        }
        by_user[t] = ku;
        by_item[t] = ki;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(kept, local);
}

// rowptr[r] = first sorted entry with row >= r (binary search); a row without entries is an error of the reference
// ("User %d has not rated any item" / "Item %d has not been rated by anybody")
__global__ void k_nmf_rowptr(int32_t n_rows, int64_t nnz, const uint64_t* __restrict__ keys, int32_t* __restrict__ rowptr,
                             int32_t* __restrict__ first_empty) {
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)(keys[mid] >> 32) < (int64_t)r) lo = mid + 1; else hi = mid;
        }
        rowptr[r] = (int32_t)lo;
    }
}
__global__ void k_nmf_check_rows(int32_t n_rows, const int32_t* __restrict__ rowptr, int32_t* __restrict__ first_empty) {
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x)
        if (rowptr[r + 1] == rowptr[r]) atomicMin(first_empty, r);
}

// SpMM  X[r][:] = sum over the row's ratings a * B[col][:]  in two deterministic stages: a row is cut into chunks of at most
// NMF_CHUNK entries (the most popular item of ML-25M shape has ~10^5 raters: one wave walking it alone took 50 ms), one
// wave per chunk writes a partial k-vector (lanes over the k columns, four entries in flight), then the row's partials are
// added in chunk order.
constexpr int NMF_CHUNK = 512;
__global__ void k_nmf_chunk_counts(int32_t n_rows, const int32_t* __restrict__ rowptr, int32_t* __restrict__ cnt) {
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += gridDim.x * blockDim.x)
        cnt[r] = r < n_rows ? (rowptr[r + 1] - rowptr[r] + NMF_CHUNK - 1) / NMF_CHUNK : 0;
}
__global__ void k_nmf_chunk_rows(int32_t n_rows, const int32_t* __restrict__ chunkptr, int32_t* __restrict__ chunk_row) {
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x)
        for (int32_t c = chunkptr[r]; c < chunkptr[r + 1]; c++) chunk_row[c] = r;
}
__global__ void k_nmf_spmm_chunks(int32_t n_chunks, int32_t k, const int32_t* __restrict__ chunk_row, const int32_t* __restrict__ chunkptr,
                                  const int32_t* __restrict__ rowptr, const uint64_t* __restrict__ keys, const float* __restrict__ val,
                                  const double* __restrict__ B, double* __restrict__ part) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int32_t ch = blockIdx.x * wpb + (threadIdx.x >> 6); ch < n_chunks; ch += gridDim.x * wpb) {
        const int32_t r = chunk_row[ch];
        const int32_t f0 = rowptr[r] + (ch - chunkptr[r]) * NMF_CHUNK, f1 = min(rowptr[r + 1], f0 + NMF_CHUNK);
        double acc[NMF_MAX_K / 64];
#pragma unroll
        for (int x = 0; x < NMF_MAX_K / 64; x++) acc[x] = 0.0;
        for (int32_t f = f0; f < f1; f += 4) {
            double a[4];
            const double* b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int32_t ff = min(f + u, f1 - 1);
                a[u] = f + u < f1 ? (double)val[ff] : 0.0;
                b[u] = B + (int64_t)(uint32_t)keys[ff] * k;
            }
#pragma unroll
            for (int x = 0; x < NMF_MAX_K / 64; x++) {
                const int c = lane + 64 * x;
                if (c < k) {
                    const double v0 = b[0][c], v1 = b[1][c], v2 = b[2][c], v3 = b[3][c];
                    acc[x] += a[0] * v0;      // entry order, like the sequential sum (a tail slot re-reads the last entry:
                    if (f + 1 < f1) acc[x] += a[1] * v1;      // skipped, so that 0 * inf cannot leak in)
                    if (f + 2 < f1) acc[x] += a[2] * v2;
                    if (f + 3 < f1) acc[x] += a[3] * v3;
                }
            }
        }
#pragma unroll
        for (int x = 0; x < NMF_MAX_K / 64; x++) {
            const int c = lane + 64 * x;
            if (c < k) part[(int64_t)ch * k + c] = acc[x];
        }
    }
}
__global__ void k_nmf_spmm_reduce(int32_t n_rows, int32_t k, const int32_t* __restrict__ chunkptr, const double* __restrict__ part,
                                  double* __restrict__ X) {
    const int64_t total = (int64_t)n_rows * k;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t r = (int32_t)(t / k), c = (int32_t)(t % k);
        double s = 0.0;
        for (int32_t ch = chunkptr[r]; ch < chunkptr[r + 1]; ch++) s += part[(int64_t)ch * k + c];
        X[t] = s;
    }
}

// C = M^T M in two stages: every block sums its contiguous row range into part[block][k*k], then a fixed-order sum
__global__ void k_nmf_gram_partial(int32_t n_rows, int32_t k, const double* __restrict__ M, double* __restrict__ part) {
    const int32_t per = (n_rows + gridDim.x - 1) / gridDim.x;
    const int32_t r0 = blockIdx.x * per, r1 = min(n_rows, r0 + per);
    for (int e = threadIdx.x; e < k * k; e += blockDim.x) {
        const int a = e / k, b = e % k;
        double s = 0.0;
        for (int32_t r = r0; r < r1; r++) s += M[(int64_t)r * k + a] * M[(int64_t)r * k + b];
        part[(int64_t)blockIdx.x * k * k + e] = s;
    }
}
__global__ void k_nmf_gram_sum(int32_t k, int32_t n_part, const double* __restrict__ part, double* __restrict__ C) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < k * k; e += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int p = 0; p < n_part; p++) s += part[(int64_t)p * k * k + e];
        C[e] = s;
    }
}

__device__ __forceinline__ double fy_clampinf(double v) { return isinf(v) ? (v > 0 ? DBL_MAX : -DBL_MAX) : v; }

// out[r][c] = m_c * x_c / (y_c + eps),  y = C m.   mode 0: HComputationReducer; 1: PPCHComputationReducer (+ optional L1
// normalisation); 2: WComputationMapper (infinities clamped)
__global__ void k_nmf_update(int32_t n_rows, int32_t k, const double* __restrict__ M, const double* __restrict__ X,
                             const double* __restrict__ C_, int mode, int normalize, double* __restrict__ out) {
    const double eps = 1e-12;   // MatrixComputationJob.java:41
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    // the k x k matrix in LDS (k <= 64: 32 KiB), padded rows against bank conflicts; larger k reads it through L2
    __shared__ double shC[64 * 65];
    const bool in_lds = k <= 64;
    if (in_lds) {
        for (int e = threadIdx.x; e < k * k; e += blockDim.x) shC[(e / k) * 65 + (e % k)] = C_[e];
        __syncthreads();
    }
    const double* __restrict__ C = in_lds ? shC : C_;
    const int ldc = in_lds ? 65 : k;
    for (int32_t r = blockIdx.x * wpb + (threadIdx.x >> 6); r < n_rows; r += gridDim.x * wpb) {
        const double* __restrict__ m = M + (int64_t)r * k;
        double y[NMF_MAX_K / 64], x[NMF_MAX_K / 64], mv[NMF_MAX_K / 64];
        double d = 0.0, e = 0.0;
#pragma unroll
        for (int q = 0; q < NMF_MAX_K / 64; q++) {
            const int c = lane + 64 * q;
            y[q] = 0.0; x[q] = 0.0; mv[q] = 0.0;
            if (c < k) {
                double s = 0.0;
                for (int a = 0; a < k; a++) s += C[(int64_t)c * ldc + a] * m[a];
                y[q] = s;
                x[q] = X[(int64_t)r * k + c];
                mv[q] = m[c];
                d += mv[q] * y[q];
                e += mv[q] * x[q];
            }
        }
        if (mode == 1) {
            for (int o = 32; o > 0; o >>= 1) { d += __shfl_xor(d, o, 64); e += __shfl_xor(e, o, 64); }
        }
        double res[NMF_MAX_K / 64], l1 = 0.0;
#pragma unroll
        for (int q = 0; q < NMF_MAX_K / 64; q++) {
            const int c = lane + 64 * q;
            res[q] = 0.0;
            if (c < k) {
                double xx = x[q], yy = y[q];
                if (mode == 1) { xx = fy_clampinf(xx + d); yy = fy_clampinf(yy + e); }
                else if (mode == 2) { xx = fy_clampinf(xx); yy = fy_clampinf(yy); }
                res[q] = mv[q] * (xx / (yy + eps));
                l1 += fabs(res[q]);
            }
        }
        if (normalize) {
            for (int o = 32; o > 0; o >>= 1) l1 += __shfl_xor(l1, o, 64);
        }
#pragma unroll
        for (int q = 0; q < NMF_MAX_K / 64; q++) {
            const int c = lane + 64 * q;
            if (c < k) out[(int64_t)r * k + c] = normalize ? res[q] / l1 : res[q];
        }
    }
}

void nmf_factorize(Context* ctx, const fy_nmf_params* prm, const fy_ratings* R, double* H_host, double* W_host, fy_stats* st) {
    if (!prm) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "params is NULL");
    const int32_t nU = prm->number_of_users, nI = prm->number_of_items, k = prm->number_of_clusters;
    if (nU <= 0 || nI <= 0 || k <= 0 || prm->number_of_iterations < 0) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "numberOfUsers / numberOfItems / numberOfClusters must be > 0");
    if (k > NMF_MAX_K) FY_FAIL(FY_ERR_UNSUPPORTED, "numberOfClusters %d exceeds the kernel limit %d", k, NMF_MAX_K);
    if (!H_host || !W_host) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "H / W are NULL");
    hipStream_t s = ctx->stream;
    const int64_t n_in = R->nnz;
    EventTimer t_total(ctx), t_prep(ctx);
    const size_t sp_total = t_total.begin();
    const size_t sp_prep = t_prep.begin();
    // ---- CSR by user and CSC by item of the kept ratings (two radix sorts; entries of a row in ascending partner id)
    DevBuf<uint64_t> ku(ctx, (size_t)std::max<int64_t>(1, n_in)), ki(ctx, (size_t)std::max<int64_t>(1, n_in)),
        ku_s(ctx, (size_t)std::max<int64_t>(1, n_in)), ki_s(ctx, (size_t)std::max<int64_t>(1, n_in));
    DevBuf<float> vu(ctx, (size_t)std::max<int64_t>(1, n_in)), vi(ctx, (size_t)std::max<int64_t>(1, n_in));
    DevBuf<unsigned long long> kept(ctx, 1);
    DevBuf<int> err(ctx, 1);
    kept.zero();
    err.zero();
    if (n_in) {
        k_nmf_keys<<<grid_for(n_in), 256, 0, s>>>(n_in, R->user.get(), R->item.get(), R->score.get(), nU, nI, ku.get(), ki.get(), kept.get(), err.get());
        FY_KERNEL_CHECK();
        auto bits = [](uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return std::max(1, b); };
        sort_pairs_u64_f32(ctx, ku.get(), ku_s.get(), const_cast<float*>(R->score.get()), vu.get(), (size_t)n_in, 64);
        sort_pairs_u64_f32(ctx, ki.get(), ki_s.get(), const_cast<float*>(R->score.get()), vi.get(), (size_t)n_in, 64);
        (void)bits;
    }
    const int64_t nnz = (int64_t)fetch(ctx, kept.get());
    if (fetch(ctx, err.get())) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "a rating's user / item id is outside [1, numberOfUsers] x [1, numberOfItems]");
    DevBuf<int32_t> uptr(ctx, (size_t)nU + 1), iptr(ctx, (size_t)nI + 1), empty(ctx, 2);
    FY_HIP(hipMemsetAsync(empty.get(), 0x7F, 2 * sizeof(int32_t), s));
    k_nmf_rowptr<<<grid_for((int64_t)nU + 1), 256, 0, s>>>(nU, nnz, ku_s.get(), uptr.get(), nullptr);
    FY_KERNEL_CHECK();
    k_nmf_rowptr<<<grid_for((int64_t)nI + 1), 256, 0, s>>>(nI, nnz, ki_s.get(), iptr.get(), nullptr);
    FY_KERNEL_CHECK();
    k_nmf_check_rows<<<grid_for(nU), 256, 0, s>>>(nU, uptr.get(), empty.get());
    FY_KERNEL_CHECK();
    k_nmf_check_rows<<<grid_for(nI), 256, 0, s>>>(nI, iptr.get(), empty.get() + 1);
    FY_KERNEL_CHECK();
    int32_t he[2];
    d2h(ctx, he, empty.get(), 2);
    sync(ctx);
    // HComputationReducer.java:50-53 / WComputationMapper.java:93-96
    if (he[0] < nU) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "User %d has not rated any item", he[0] + 1);
    if (he[1] < nI) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "Item %d has not been rated by anybody", he[1] + 1);
    // chunk tables of both SpMMs (fixed across the iterations)
    struct Chunks { DevBuf<int32_t> ptr, row; int32_t n = 0; };
    auto make_chunks = [&](int32_t n_rows, const int32_t* rowptr, Chunks& ck) {
        DevBuf<int32_t> cnt(ctx, (size_t)n_rows + 1);
        ck.ptr.alloc(ctx, (size_t)n_rows + 1);
        k_nmf_chunk_counts<<<grid_for((int64_t)n_rows + 1), 256, 0, s>>>(n_rows, rowptr, cnt.get());
        FY_KERNEL_CHECK();
        exclusive_scan_i32(ctx, cnt.get(), ck.ptr.get(), (size_t)n_rows + 1);
        ck.n = fetch(ctx, ck.ptr.get() + n_rows);
        ck.row.alloc(ctx, (size_t)std::max(1, ck.n));
        k_nmf_chunk_rows<<<grid_for(n_rows), 256, 0, s>>>(n_rows, ck.ptr.get(), ck.row.get());
        FY_KERNEL_CHECK();
    };
    Chunks cu, ci;
    make_chunks(nU, uptr.get(), cu);
    make_chunks(nI, iptr.get(), ci);
    t_prep.end(sp_prep);

    DevBuf<double> H(ctx, (size_t)nU * k), W(ctx, (size_t)nI * k), H2(ctx, (size_t)nU * k), W2(ctx, (size_t)nI * k);
    DevBuf<double> XH(ctx, (size_t)nU * k), XW(ctx, (size_t)nI * k), C(ctx, (size_t)k * k), part(ctx, (size_t)NMF_GRAM_BLOCKS * k * k);
    DevBuf<double> spart(ctx, (size_t)std::max(cu.n, ci.n) * k + 1);
    h2d(ctx, H.get(), H_host, (size_t)nU * k);
    h2d(ctx, W.get(), W_host, (size_t)nI * k);
    double* h = H.get();
    double* w = W.get();
    double* h2 = H2.get();
    double* w2 = W2.get();
    const int f = prm->normalization_frequency;
    EventTimer t_iter(ctx);      // the iterations alone: H / W are in HBM (the uploads are in front of it, the downloads behind)
    const size_t sp_iter = t_iter.begin();
    for (int32_t it = 1; it <= prm->number_of_iterations; it++) {
        // H2 from (H, W)
        k_nmf_spmm_chunks<<<grid_for((int64_t)cu.n * 64, 256), 256, 0, s>>>(cu.n, k, cu.row.get(), cu.ptr.get(), uptr.get(), ku_s.get(), vu.get(), w, spart.get());
        FY_KERNEL_CHECK();
        k_nmf_spmm_reduce<<<grid_for((int64_t)nU * k), 256, 0, s>>>(nU, k, cu.ptr.get(), spart.get(), XH.get());
        FY_KERNEL_CHECK();
        k_nmf_gram_partial<<<NMF_GRAM_BLOCKS, 256, 0, s>>>(nI, k, w, part.get());
        FY_KERNEL_CHECK();
        k_nmf_gram_sum<<<grid_for((int64_t)k * k), 256, 0, s>>>(k, NMF_GRAM_BLOCKS, part.get(), C.get());
        FY_KERNEL_CHECK();
        const int normalize = prm->ppc && f != 0 && (it % f == 0);   // Java %: f = -1 (the key left unset) normalises every iteration
        k_nmf_update<<<grid_for((int64_t)nU * 64, 256), 256, 0, s>>>(nU, k, h, XH.get(), C.get(), prm->ppc ? 1 : 0, normalize, h2);
        FY_KERNEL_CHECK();
        // W2 from the same (H, W)
        k_nmf_spmm_chunks<<<grid_for((int64_t)ci.n * 64, 256), 256, 0, s>>>(ci.n, k, ci.row.get(), ci.ptr.get(), iptr.get(), ki_s.get(), vi.get(), h, spart.get());
        FY_KERNEL_CHECK();
        k_nmf_spmm_reduce<<<grid_for((int64_t)nI * k), 256, 0, s>>>(nI, k, ci.ptr.get(), spart.get(), XW.get());
        FY_KERNEL_CHECK();
        k_nmf_gram_partial<<<NMF_GRAM_BLOCKS, 256, 0, s>>>(nU, k, h, part.get());
        FY_KERNEL_CHECK();
        k_nmf_gram_sum<<<grid_for((int64_t)k * k), 256, 0, s>>>(k, NMF_GRAM_BLOCKS, part.get(), C.get());
        FY_KERNEL_CHECK();
        k_nmf_update<<<grid_for((int64_t)nI * 64, 256), 256, 0, s>>>(nI, k, w, XW.get(), C.get(), 2, 0, w2);
        FY_KERNEL_CHECK();
        std::swap(h, h2);
        std::swap(w, w2);
    }
    t_iter.end(sp_iter);
    d2h(ctx, H_host, h, (size_t)nU * k);
    d2h(ctx, W_host, w, (size_t)nI * k);
    t_total.end(sp_total);
    sync(ctx);
    if (st) {
        *st = fy_stats{};
        st->nnz = nnz;
        st->n_users = nU;
        st->n_items = nI;
        st->ms_prepare = t_prep.total_ms();
        st->ms_total = t_total.total_ms();
        st->ms_cooc = t_iter.total_ms();      // (fy_stats has no field of its own for it: "the job's main kernels", like the RM2 matrix build)
    }
}

}  // namespace fy
